/* tests/standin/standin_transcript.c -- TEST / BENCH SCAFFOLDING, not part of the product and not part of the oracle.
 *
 * A native stand-in for the reference's Keccak transcript (src/provider/keccak.rs, host code that stays in Rust): the sum-check
 * provers -- the product's nmx_sumcheck_prove_* and the oracle's ref_sumcheck_prove_* alike -- call back into the transcript once
 * per round (`absorb(b"p", &poly); squeeze(b"c")`).  A Python callback costs 10-20 us per round, more than a round's GPU time, so
 * the timed replays use this one: a chained 256-bit mix of everything absorbed, squeezed to 31 bytes (< 2^248 < p for all four
 * fields, so every challenge is a canonical element).  Deterministic: two provers fed the same inputs see the same challenges iff
 * their round polynomials agree.  NOT cryptographic. */
#include <stddef.h>
#include <stdint.h>
#include <string.h>

typedef struct { uint64_t s[4]; uint64_t absorbed, squeezed; } standin_t;

static uint64_t mix64(uint64_t x) { /* splitmix64 finaliser */
  x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull; x ^= x >> 27; x *= 0x94d049bb133111ebull; x ^= x >> 31; return x;
}
void standin_init(standin_t* t, uint64_t seed) {
  for (int i = 0; i < 4; i++) t->s[i] = mix64(seed + 0x9e3779b97f4a7c15ull * (uint64_t)(i + 1));
  t->absorbed = t->squeezed = 0;
}
void standin_absorb(standin_t* t, const uint8_t* data, size_t n) {
  for (size_t i = 0; i < n; i += 8) {
    uint64_t w = 0;
    memcpy(&w, data + i, n - i < 8 ? n - i : 8);
    const int j = (int)((t->absorbed++) & 3);
    t->s[j] = mix64(t->s[j] ^ w) + t->s[(j + 1) & 3];
  }
}
void standin_squeeze(standin_t* t, uint8_t out[32]) {
  uint64_t o[4];
  for (int i = 0; i < 4; i++) {
    t->s[i] = mix64(t->s[i] + t->s[(i + 3) & 3] + (++t->squeezed));
    o[i] = t->s[i];
  }
  memcpy(out, o, 32);
  out[31] = 0; /* < 2^248 */
}
/* nmx_transcript_fn / ref_transcript_fn */
int standin_transcript(void* ctx, const uint8_t* coeffs, size_t n_coeffs, uint8_t* challenge32) {
  standin_t* t = (standin_t*)ctx;
  standin_absorb(t, coeffs, 32 * n_coeffs);
  standin_squeeze(t, challenge32);
  return 0;
}
/* nmx_ipa_transcript_fn / ref_ipa_transcript_fn: absorb(b"L", &L); absorb(b"R", &R); squeeze(b"r") (src/provider/ipa_pc.rs:231-234) */
int standin_ipa_transcript(void* ctx, const uint8_t* L_xy64, int L_is_inf, const uint8_t* R_xy64, int R_is_inf, uint8_t* r32) {
  standin_t* t = (standin_t*)ctx;
  const uint8_t flags[8] = {(uint8_t)(L_is_inf != 0), (uint8_t)(R_is_inf != 0), 0, 0, 0, 0, 0, 0};
  standin_absorb(t, L_xy64, 64);
  standin_absorb(t, R_xy64, 64);
  standin_absorb(t, flags, 8);
  standin_squeeze(t, r32);
  r32[0] |= 1; /* never zero: the reference unwraps the inverse (:235) */
  return 0;
}
