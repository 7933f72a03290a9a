/* oracle/nova_ref.c -- tier-2 oracle: CPU restatement of the reference's MSM provider, in plain C.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may load
 * libnova_ref.so; nothing under nova_amd/ links or calls it.  It doubles as the timed CPU baseline
 * (`cpu_baseline.kind = "port"`): the Rust reference (halo2curves asm + rayon) cannot be built in this
 * environment (no cargo/rustc, crates not vendored -- SURVEY.md section 0).
 *
 * What is restated (paths relative to /root/reference):
 *   src/provider/msm.rs:38-183    BucketXYZZ: zero / double_in_place (dbl-2008-s-1) / add_assign_bucket
 *                                 (add-2008-s) / bucket_add_affine (madd-2008-s) incl. exceptional cases
 *   src/provider/msm.rs:189-211   scalar_num_bits, repr_low_u64
 *   src/provider/msm.rs:225-419   msm(): n == 0, n <= 16 -> msm_simple, zero/identity filtering, signed
 *                                 classification into 11 bit-width groups, per-group algorithms, pos - neg
 *   src/provider/msm.rs:432-454   accumulate_bases (chunked sum)
 *   src/provider/msm.rs:478-503   msm_small_with_max_num_bits dispatch (0 / 1 / 2..=10 / 11..=32 / else)
 *   src/provider/msm.rs:505-530   msm_binary        :533-575 msm_10        :577-677 msm_small_rest
 *   src/provider/msm.rs:679-686   compute_ln        :689-708 batch_add
 *   src/provider/traits.rs:82-90  batch_vartime_multiscalar_mul (bases[..len_j] per vector)
 *   src/provider/pedersen.rs:263-270 / hyperkzg.rs:584-591  commit = msm + h*r
 *   src/provider/traits.rs:303-312 to_coordinates(): identity -> (0, 0, true)
 * Third-party piece: `halo2curves::msm::msm_best` (crate halo2curves = "0.9.0", Cargo.toml:36-41; called at
 * msm.rs:411,500) is not in the tree.  Its published algorithm is a windowed Pippenger over rayon chunks
 * (window c = 1 if n < 4, 3 if n < 32, else ceil(ln n); signed "Booth" digits; per-window buckets summed by a
 * running sum; windows combined by c doublings); `best_msm()` below restates that.  Any correct MSM returns the
 * same group element, so the internal strategy affects the baseline's speed only.
 *
 * PARITY UNPINNED at reference-output level: this oracle has never met an output of the reference itself (no Rust toolchain in the
 * image, probed again 2026-09-30; no C / C++ reference sources to build into oracle/_ref), and the reference stores no MSM output
 * vectors (its tests assert msm == naive sum at run time: msm.rs:722-821, curve_property_tests.rs:180-218, blitzar.rs:48-214).
 * What pins it instead: oracle/pyref.py (big-int definition) on that same test matrix (tests/test_oracle.py), the public EIP-196
 * vectors and the SymPy-computed vectors of all four curves (tests/golden/public_kats.json, sympy_kats.json), and -- for the
 * field-vector and sum-check rows -- every known answer the reference's own tests hold (tests/golden/field_kats.json, with their
 * source lines) plus the reference's verifier equations on every replayed proof.
 *
 * Threading: OpenMP, one chunk per thread + final sum -- the decomposition the reference uses with rayon
 * (`par_chunks(len / num_threads)` + `reduce(identity, +)`, msm.rs:520-526,564-571,664-673).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef unsigned __int128 u128;
typedef struct { uint64_t l[4]; } fe;

typedef struct {
  fe p;          /* modulus */
  uint64_t ninv; /* -p^-1 mod 2^64 */
  fe r1;         /* R mod p (Montgomery one) */
  fe r2;         /* R^2 mod p */
} field_t;

typedef struct {
  const field_t* base;   /* coordinate field */
  const field_t* scalar; /* scalar field */
} curve_t;

/* moduli: bn256_grumpkin.rs:39-40, pasta.rs:37-38; Montgomery constants computed from them (tests re-derive) */
static const field_t F_BN_Q = {
    {{0x3c208c16d87cfd47ull, 0x97816a916871ca8dull, 0xb85045b68181585dull, 0x30644e72e131a029ull}},
    0x87d20782e4866389ull,
    {{0xd35d438dc58f0d9dull, 0x0a78eb28f5c70b3dull, 0x666ea36f7879462cull, 0x0e0a77c19a07df2full}},
    {{0xf32cfc5b538afa89ull, 0xb5e71911d44501fbull, 0x47ab1eff0a417ff6ull, 0x06d89f71cab8351full}}};
static const field_t F_BN_R = {
    {{0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull}},
    0xc2e1f593efffffffull,
    {{0xac96341c4ffffffbull, 0x36fc76959f60cd29ull, 0x666ea36f7879462eull, 0x0e0a77c19a07df2full}},
    {{0x1bb8e645ae216da7ull, 0x53fe3ab1e35c59e3ull, 0x8c49833d53bb8085ull, 0x0216d0b17f4e44a5ull}}};
static const field_t F_PA_P = {
    {{0x992d30ed00000001ull, 0x224698fc094cf91bull, 0x0000000000000000ull, 0x4000000000000000ull}},
    0x992d30ecffffffffull,
    {{0x34786d38fffffffdull, 0x992c350be41914adull, 0xffffffffffffffffull, 0x3fffffffffffffffull}},
    {{0x8c78ecb30000000full, 0xd7d30dbd8b0de0e7ull, 0x7797a99bc3c95d18ull, 0x096d41af7b9cb714ull}}};
static const field_t F_PA_Q = {
    {{0x8c46eb2100000001ull, 0x224698fc0994a8ddull, 0x0000000000000000ull, 0x4000000000000000ull}},
    0x8c46eb20ffffffffull,
    {{0x5b2b3e9cfffffffdull, 0x992c350be3420567ull, 0xffffffffffffffffull, 0x3fffffffffffffffull}},
    {{0xfc9678ff0000000full, 0x67bb433d891a16e3ull, 0x7fae231004ccf590ull, 0x096d41af7ccfdaa9ull}}};

/* curve ids as include/nova_mi355x.h: 0 bn254 g1, 1 grumpkin, 2 pallas, 3 vesta */
static const curve_t CURVES[4] = {{&F_BN_Q, &F_BN_R}, {&F_BN_R, &F_BN_Q}, {&F_PA_P, &F_PA_Q}, {&F_PA_Q, &F_PA_P}};

/* ------------------------------------------------------------------ field ------------------------------ */
static inline int fe_is_zero(const fe* a) { return (a->l[0] | a->l[1] | a->l[2] | a->l[3]) == 0; }
static inline int fe_eq(const fe* a, const fe* b) {
  return ((a->l[0] ^ b->l[0]) | (a->l[1] ^ b->l[1]) | (a->l[2] ^ b->l[2]) | (a->l[3] ^ b->l[3])) == 0;
}
static inline int fe_geq(const fe* a, const fe* b) {
  for (int i = 3; i >= 0; i--) {
    if (a->l[i] > b->l[i]) return 1;
    if (a->l[i] < b->l[i]) return 0;
  }
  return 1;
}
static inline void fe_sub_raw(fe* r, const fe* a, const fe* b, uint64_t* borrow) {
  u128 bw = 0;
  for (int i = 0; i < 4; i++) {
    u128 d = (u128)a->l[i] - b->l[i] - bw;
    r->l[i] = (uint64_t)d;
    bw = (d >> 64) & 1;
  }
  *borrow = (uint64_t)bw;
}
static inline void fe_add(const field_t* F, fe* r, const fe* a, const fe* b) {
  u128 c = 0;
  fe t;
  for (int i = 0; i < 4; i++) {
    c += (u128)a->l[i] + b->l[i];
    t.l[i] = (uint64_t)c;
    c >>= 64;
  }
  uint64_t bw;
  fe u;
  fe_sub_raw(&u, &t, &F->p, &bw);
  *r = bw ? t : u; /* p < 2^255: no carry out of the top limb */
}
static inline void fe_sub(const field_t* F, fe* r, const fe* a, const fe* b) {
  uint64_t bw;
  fe t;
  fe_sub_raw(&t, a, b, &bw);
  if (bw) {
    u128 c = 0;
    for (int i = 0; i < 4; i++) {
      c += (u128)t.l[i] + F->p.l[i];
      t.l[i] = (uint64_t)c;
      c >>= 64;
    }
  }
  *r = t;
}
static inline void fe_neg(const field_t* F, fe* r, const fe* a) {
  if (fe_is_zero(a)) { *r = *a; return; }
  uint64_t bw;
  fe_sub_raw(r, &F->p, a, &bw);
}
static inline void fe_dbl(const field_t* F, fe* r, const fe* a) { fe_add(F, r, a, a); }
/* Montgomery product, CIOS with 64-bit limbs */
static inline void fe_mul(const field_t* F, fe* r, const fe* a, const fe* b) {
  uint64_t t[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 4; i++) {
    u128 c = 0;
    for (int j = 0; j < 4; j++) {
      c += (u128)a->l[j] * b->l[i] + t[j];
      t[j] = (uint64_t)c;
      c >>= 64;
    }
    c += t[4];
    t[4] = (uint64_t)c;
    t[5] = (uint64_t)(c >> 64);
    uint64_t m = t[0] * F->ninv;
    c = (u128)m * F->p.l[0] + t[0];
    c >>= 64;
    for (int j = 1; j < 4; j++) {
      c += (u128)m * F->p.l[j] + t[j];
      t[j - 1] = (uint64_t)c;
      c >>= 64;
    }
    c += t[4];
    t[3] = (uint64_t)c;
    t[4] = t[5] + (uint64_t)(c >> 64);
  }
  fe x = {{t[0], t[1], t[2], t[3]}};
  uint64_t bw;
  fe u;
  fe_sub_raw(&u, &x, &F->p, &bw);
  *r = (t[4] || !bw) ? u : x;
}
static inline void fe_sqr(const field_t* F, fe* r, const fe* a) { fe_mul(F, r, a, a); }
static void fe_inv(const field_t* F, fe* r, const fe* a) { /* a^(p-2) */
  fe e = F->p;
  u128 bw = 2;
  for (int i = 0; i < 4; i++) {
    u128 d = (u128)e.l[i] - bw;
    e.l[i] = (uint64_t)d;
    bw = (d >> 64) & 1;
  }
  fe acc = F->r1;
  for (int i = 255; i >= 0; i--) {
    fe_sqr(F, &acc, &acc);
    if ((e.l[i >> 6] >> (i & 63)) & 1) fe_mul(F, &acc, &acc, a);
  }
  *r = acc;
}
static inline void fe_to_mont(const field_t* F, fe* r, const fe* a) { fe_mul(F, r, a, &F->r2); }
static inline void fe_from_mont(const field_t* F, fe* r, const fe* a) {
  fe one = {{1, 0, 0, 0}};
  fe_mul(F, r, a, &one);
}

/* ------------------------------------------------------------------ XYZZ (msm.rs:38-183) ---------------- */
typedef struct { fe x, y; } aff;            /* identity = (0,0) */
typedef struct { fe x, y, zz, zzz; } xyzz;  /* identity <=> zz == 0 (msm.rs:59-61) */

static inline int aff_is_identity(const aff* p) { return fe_is_zero(&p->x) && fe_is_zero(&p->y); }
static inline void xyzz_zero(const field_t* F, xyzz* b) { /* msm.rs:48-55 */
  b->x = F->r1; b->y = F->r1;
  memset(&b->zz, 0, sizeof(fe)); memset(&b->zzz, 0, sizeof(fe));
}
static inline int xyzz_is_zero(const xyzz* b) { return fe_is_zero(&b->zz); }

static void xyzz_double(const field_t* F, xyzz* b) { /* msm.rs:65-88 */
  if (xyzz_is_zero(b)) return;
  fe u, v, w, s, xx, m, t, x3, y3;
  fe_dbl(F, &u, &b->y);
  fe_sqr(F, &v, &u);
  fe_mul(F, &w, &u, &v);
  fe_mul(F, &s, &b->x, &v);
  fe_sqr(F, &xx, &b->x);
  fe_dbl(F, &m, &xx); fe_add(F, &m, &m, &xx);
  fe_sqr(F, &x3, &m); fe_dbl(F, &t, &s); fe_sub(F, &x3, &x3, &t);
  fe_sub(F, &t, &s, &x3); fe_mul(F, &y3, &m, &t);
  fe_mul(F, &t, &w, &b->y); fe_sub(F, &y3, &y3, &t);
  b->x = x3; b->y = y3;
  fe_mul(F, &b->zz, &b->zz, &v);
  fe_mul(F, &b->zzz, &b->zzz, &w);
}
static void xyzz_add(const field_t* F, xyzz* a, const xyzz* o) { /* msm.rs:91-123 */
  if (xyzz_is_zero(o)) return;
  if (xyzz_is_zero(a)) { *a = *o; return; }
  fe u1, u2, s1, s2;
  fe_mul(F, &u1, &a->x, &o->zz);
  fe_mul(F, &u2, &o->x, &a->zz);
  fe_mul(F, &s1, &a->y, &o->zzz);
  fe_mul(F, &s2, &o->y, &a->zzz);
  if (fe_eq(&u1, &u2)) {
    if (fe_eq(&s1, &s2)) xyzz_double(F, a); else xyzz_zero(F, a);
    return;
  }
  fe p, r, pp, ppp, q, t, x3;
  fe_sub(F, &p, &u2, &u1);
  fe_sub(F, &r, &s2, &s1);
  fe_sqr(F, &pp, &p);
  fe_mul(F, &ppp, &p, &pp);
  fe_mul(F, &q, &u1, &pp);
  fe_sqr(F, &x3, &r); fe_sub(F, &x3, &x3, &ppp); fe_dbl(F, &t, &q); fe_sub(F, &x3, &x3, &t);
  fe_sub(F, &t, &q, &x3); fe_mul(F, &t, &r, &t);
  fe_mul(F, &s1, &s1, &ppp); fe_sub(F, &a->y, &t, &s1);
  a->x = x3;
  fe_mul(F, &a->zz, &a->zz, &o->zz); fe_mul(F, &a->zz, &a->zz, &pp);
  fe_mul(F, &a->zzz, &a->zzz, &o->zzz); fe_mul(F, &a->zzz, &a->zzz, &ppp);
}
static void xyzz_add_affine(const field_t* F, xyzz* b, const aff* p) { /* msm.rs:129-165 */
  if (aff_is_identity(p)) return;
  if (xyzz_is_zero(b)) { b->x = p->x; b->y = p->y; b->zz = F->r1; b->zzz = F->r1; return; }
  fe u2, s2;
  fe_mul(F, &u2, &p->x, &b->zz);
  fe_mul(F, &s2, &p->y, &b->zzz);
  if (fe_eq(&b->x, &u2)) {
    if (fe_eq(&b->y, &s2)) xyzz_double(F, b); else xyzz_zero(F, b);
    return;
  }
  fe pv, r, pp, ppp, q, t, x3;
  fe_sub(F, &pv, &u2, &b->x);
  fe_sub(F, &r, &s2, &b->y);
  fe_sqr(F, &pp, &pv);
  fe_mul(F, &ppp, &pv, &pp);
  fe_mul(F, &q, &b->x, &pp);
  fe_sqr(F, &x3, &r); fe_sub(F, &x3, &x3, &ppp); fe_dbl(F, &t, &q); fe_sub(F, &x3, &x3, &t);
  fe_sub(F, &t, &q, &x3); fe_mul(F, &t, &r, &t);
  fe_mul(F, &q, &b->y, &ppp); fe_sub(F, &b->y, &t, &q);
  b->x = x3;
  fe_mul(F, &b->zz, &b->zz, &pp);
  fe_mul(F, &b->zzz, &b->zzz, &ppp);
}
static void xyzz_neg(const field_t* F, xyzz* a) { fe_neg(F, &a->y, &a->y); }
static void xyzz_sub(const field_t* F, xyzz* a, const xyzz* o) {
  xyzz t = *o; xyzz_neg(F, &t); xyzz_add(F, a, &t);
}
static void xyzz_to_affine(const field_t* F, aff* r, const xyzz* b) { /* msm.rs:172-183 + traits.rs:303-312 */
  if (xyzz_is_zero(b)) { memset(r, 0, sizeof(*r)); return; }
  fe zi, zzi;
  fe_inv(F, &zi, &b->zz);
  fe_inv(F, &zzi, &b->zzz);
  fe_mul(F, &r->x, &b->x, &zi);
  fe_mul(F, &r->y, &b->y, &zzi);
}
/* k * P, k a canonical 256-bit integer (msm_simple's `*base * coeff`, msm.rs:422-429) */
static void xyzz_scalar_mul(const field_t* F, xyzz* r, const aff* p, const fe* k) {
  xyzz acc; xyzz_zero(F, &acc);
  for (int i = 255; i >= 0; i--) {
    xyzz_double(F, &acc);
    if ((k->l[i >> 6] >> (i & 63)) & 1) xyzz_add_affine(F, &acc, p);
  }
  *r = acc;
}

/* ------------------------------------------------------------------ helpers ----------------------------- */
static int g_threads = 0;
static int nthreads(void) {
  if (g_threads > 0) return g_threads;
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
static uint32_t scalar_num_bits(const fe* s) { /* msm.rs:191-200 (canonical integer) */
  for (int i = 3; i >= 0; i--)
    if (s->l[i]) return (uint32_t)(i * 64 + 64 - __builtin_clzll(s->l[i]));
  return 0;
}
static size_t num_bits_usize(uint64_t n) { return n == 0 ? 0 : (size_t)(64 - __builtin_clzll(n)); } /* msm.rs:456-462 */
static size_t compute_ln(size_t a) { return a == 0 ? 0 : (size_t)(63 - __builtin_clzll((uint64_t)a)) * 69 / 100; } /* msm.rs:679-686 */

/* sum of chunk results: rayon `.reduce(identity, +)` */
typedef void (*chunk_fn)(const curve_t* C, const void* scalars, const aff* bases, size_t lo, size_t hi, void* ctx,
                         xyzz* out);
static void par_chunks(const curve_t* C, const void* scalars, const aff* bases, size_t n, size_t chunk, chunk_fn fn,
                       void* ctx, xyzz* out) {
  const field_t* F = C->base;
  size_t nchunks = (n + chunk - 1) / chunk;
  xyzz* parts = (xyzz*)malloc(sizeof(xyzz) * (nchunks ? nchunks : 1));
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads())
  for (long ci = 0; ci < (long)nchunks; ci++) {
    size_t lo = (size_t)ci * chunk, hi = lo + chunk < n ? lo + chunk : n;
    fn(C, scalars, bases, lo, hi, ctx, &parts[ci]);
  }
  xyzz_zero(F, out);
  for (size_t i = 0; i < nchunks; i++) xyzz_add(F, out, &parts[i]);
  free(parts);
}

/* ------------------------------------------------------------------ small-scalar MSMs ------------------- */
/* msm_binary (msm.rs:505-530): every non-zero scalar counts as 1 */
static void binary_chunk(const curve_t* C, const void* sv, const aff* bases, size_t lo, size_t hi, void* ctx, xyzz* out) {
  (void)ctx;
  const uint64_t* s = (const uint64_t*)sv;
  xyzz_zero(C->base, out);
  for (size_t i = lo; i < hi; i++)
    if (s[i] != 0) xyzz_add_affine(C->base, out, &bases[i]);
}
static void msm_binary(const curve_t* C, const uint64_t* s, const aff* bases, size_t n, xyzz* out) {
  size_t nt = (size_t)nthreads();
  if (n > nt) par_chunks(C, s, bases, n, n / nt, binary_chunk, NULL, out);
  else binary_chunk(C, s, bases, 0, n, NULL, out);
}
/* msm_10 (msm.rs:533-575) */
static void msm10_chunk(const curve_t* C, const void* sv, const aff* bases, size_t lo, size_t hi, void* ctx, xyzz* out) {
  const field_t* F = C->base;
  const uint64_t* s = (const uint64_t*)sv;
  size_t max_num_bits = *(size_t*)ctx;
  size_t nb = (size_t)1 << max_num_bits;
  xyzz* buckets = (xyzz*)malloc(sizeof(xyzz) * nb);
  for (size_t k = 0; k < nb; k++) xyzz_zero(F, &buckets[k]);
  for (size_t i = lo; i < hi; i++)
    if (s[i] != 0) xyzz_add_affine(F, &buckets[s[i]], &bases[i]); /* in-contract: s[i] < 2^max_num_bits */
  xyzz result, running;
  xyzz_zero(F, &result); xyzz_zero(F, &running);
  for (size_t k = nb - 1; k >= 1; k--) { /* buckets.skip(1).rev() */
    xyzz_add(F, &running, &buckets[k]);
    xyzz_add(F, &result, &running);
  }
  free(buckets);
  *out = result;
}
static void msm_10(const curve_t* C, const uint64_t* s, const aff* bases, size_t n, size_t max_num_bits, xyzz* out) {
  size_t nt = (size_t)nthreads();
  if (n > nt) par_chunks(C, s, bases, n, n / nt, msm10_chunk, &max_num_bits, out);
  else msm10_chunk(C, s, bases, 0, n, &max_num_bits, out);
}
/* msm_small_rest (msm.rs:577-677) */
static void rest_chunk(const curve_t* C, const void* sv, const aff* bases, size_t lo, size_t hi, void* ctx, xyzz* out) {
  const field_t* F = C->base;
  const uint64_t* s = (const uint64_t*)sv;
  size_t max_num_bits = *(size_t*)ctx;
  size_t len = hi - lo;
  size_t c = len < 32 ? 3 : compute_ln(len) + 2;               /* msm.rs:587-591 */
  if (max_num_bits == 32 || max_num_bits == 64) c = 8;         /* msm.rs:593-595 */
  size_t nwin = (max_num_bits + c - 1) / c;                    /* (0..max_num_bits).step_by(c) */
  size_t nb = ((size_t)1 << c) - 1;
  xyzz* wsum = (xyzz*)malloc(sizeof(xyzz) * (nwin ? nwin : 1));
  xyzz* buckets = (xyzz*)malloc(sizeof(xyzz) * nb);
  for (size_t w = 0; w < nwin; w++) {
    size_t w_start = w * c;
    xyzz res; xyzz_zero(F, &res);
    for (size_t k = 0; k < nb; k++) xyzz_zero(F, &buckets[k]);
    for (size_t i = lo; i < hi; i++) {
      uint64_t sc = s[i];
      if (sc == 0) continue;
      if (sc == 1) {                                           /* msm.rs:613-617 */
        if (w_start == 0) xyzz_add_affine(F, &res, &bases[i]);
      } else {
        sc >>= w_start;
        sc %= ((uint64_t)1 << c);
        if (sc != 0) xyzz_add_affine(F, &buckets[sc - 1], &bases[i]);
      }
    }
    xyzz running; xyzz_zero(F, &running);
    for (size_t k = nb; k-- > 0;) {                            /* msm.rs:638-642 */
      xyzz_add(F, &running, &buckets[k]);
      xyzz_add(F, &res, &running);
    }
    wsum[w] = res;
  }
  /* lowest + fold(rev(window_sums[1..])) with c doublings (msm.rs:648-661) */
  xyzz total; xyzz_zero(F, &total);
  for (size_t w = nwin; w-- > 1;) {
    xyzz_add(F, &total, &wsum[w]);
    for (size_t q = 0; q < c; q++) xyzz_double(F, &total);
  }
  xyzz_add(F, &total, &wsum[0]);
  free(buckets); free(wsum);
  *out = total;
}
static void msm_small_rest(const curve_t* C, const uint64_t* s, const aff* bases, size_t n, size_t max_num_bits, xyzz* out) {
  size_t nt = (size_t)nthreads();
  if (n > nt) par_chunks(C, s, bases, n, n / nt, rest_chunk, &max_num_bits, out);
  else rest_chunk(C, s, bases, 0, n, &max_num_bits, out);
}

/* ------------------------------------------------------------------ msm_best role ----------------------- */
/* Windowed Pippenger with signed (Booth) digits over canonical 256-bit scalars; one serial instance per chunk. */
static void best_chunk(const curve_t* C, const void* sv, const aff* bases, size_t lo, size_t hi, void* ctx, xyzz* out) {
  (void)ctx;
  const field_t* F = C->base;
  const fe* s = (const fe*)sv;
  size_t len = hi - lo;
  size_t c;
  if (len < 4) c = 1;
  else if (len < 32) c = 3;
  else c = (size_t)ceil(log((double)len));
  size_t nwin = (256 + c - 1) / c + 1; /* +1: Booth carry */
  size_t nb = (size_t)1 << (c - 1);
  xyzz* buckets = (xyzz*)malloc(sizeof(xyzz) * nb);
  xyzz acc; xyzz_zero(F, &acc);
  for (size_t w = nwin; w-- > 0;) {
    for (size_t q = 0; q < c; q++) xyzz_double(F, &acc);
    for (size_t k = 0; k < nb; k++) xyzz_zero(F, &buckets[k]);
    for (size_t i = lo; i < hi; i++) {
      /* Booth digit of window w: bits [w*c - 1, w*c + c) of the scalar, d = ((v + 1) >> 1) - sign * 2^c */
      size_t bit = w * c;
      uint64_t v = 0;
      for (size_t b = 0; b <= c; b++) {
        long pos = (long)bit - 1 + (long)b;
        uint64_t bv = (pos < 0 || pos >= 256) ? 0 : ((s[i].l[pos >> 6] >> (pos & 63)) & 1);
        v |= bv << b;
      }
      int sign = (int)((v >> c) & 1);
      long d = (long)((v + 1) >> 1);
      if (sign) d -= (long)1 << c;
      if (d == 0) continue;
      aff p = bases[i];
      if (aff_is_identity(&p)) continue;
      if (d < 0) { fe_neg(F, &p.y, &p.y); d = -d; }
      xyzz_add_affine(F, &buckets[d - 1], &p);
    }
    xyzz running, sum; xyzz_zero(F, &running); xyzz_zero(F, &sum);
    for (size_t k = nb; k-- > 0;) {
      xyzz_add(F, &running, &buckets[k]);
      xyzz_add(F, &sum, &running);
    }
    xyzz_add(F, &acc, &sum);
  }
  free(buckets);
  *out = acc;
}
static void best_msm(const curve_t* C, const fe* s, const aff* bases, size_t n, xyzz* out) {
  size_t nt = (size_t)nthreads();
  if (n == 0) { xyzz_zero(C->base, out); return; }
  size_t chunk = (n + nt - 1) / nt;
  par_chunks(C, s, bases, n, chunk, best_chunk, NULL, out);
}

/* msm_small_with_max_num_bits (msm.rs:478-503) */
static void msm_small_bits(const curve_t* C, const uint64_t* s, const aff* bases, size_t n, size_t max_num_bits, xyzz* out) {
  if (max_num_bits == 0) { xyzz_zero(C->base, out); return; }
  if (max_num_bits == 1) { msm_binary(C, s, bases, n, out); return; }
  if (max_num_bits <= 10) { msm_10(C, s, bases, n, max_num_bits, out); return; }
  if (max_num_bits <= 32) { msm_small_rest(C, s, bases, n, max_num_bits, out); return; }
  fe* fs = (fe*)calloc(n ? n : 1, sizeof(fe));                  /* Scalar::from(u64), kept canonical */
  for (size_t i = 0; i < n; i++) fs[i].l[0] = s[i];
  best_msm(C, fs, bases, n, out);
  free(fs);
}

/* accumulate_bases (msm.rs:432-454) */
static void accum_chunk(const curve_t* C, const void* sv, const aff* bases, size_t lo, size_t hi, void* ctx, xyzz* out) {
  (void)sv; (void)ctx;
  xyzz_zero(C->base, out);
  for (size_t i = lo; i < hi; i++) xyzz_add_affine(C->base, out, &bases[i]);
}
static void accumulate_bases(const curve_t* C, const aff* bases, size_t n, xyzz* out) {
  size_t nt = (size_t)nthreads();
  if (n == 0) { xyzz_zero(C->base, out); return; }
  if (n > nt) par_chunks(C, NULL, bases, n, (n + nt - 1) / nt, accum_chunk, NULL, out);
  else accum_chunk(C, NULL, bases, 0, n, NULL, out);
}

/* ------------------------------------------------------------------ msm() (msm.rs:225-419) -------------- */
static void msm_full(const curve_t* C, const fe* coeffs /* canonical */, const aff* bases /* Montgomery */, size_t n, xyzz* out) {
  const field_t* F = C->base;
  const field_t* S = C->scalar;
  xyzz_zero(F, out);
  if (n == 0) return;                                              /* :228 */
  if (n <= 16) {                                                   /* :233 msm_simple */
    for (size_t i = 0; i < n; i++) {
      xyzz t; xyzz_scalar_mul(F, &t, &bases[i], &coeffs[i]);
      xyzz_add(F, out, &t);
    }
    return;
  }
  /* Phase 1: classify (:243-279) */
  uint8_t* group = (uint8_t*)malloc(n);
  size_t counts[12]; memset(counts, 0, sizeof(counts));
#pragma omp parallel for num_threads(nthreads())
  for (long i = 0; i < (long)n; i++) {
    const fe* s = &coeffs[i];
    if (fe_is_zero(s) || aff_is_identity(&bases[i])) { group[i] = 255; continue; }
    fe neg; uint64_t bw; fe_sub_raw(&neg, &S->p, s, &bw);          /* -s, canonical */
    uint32_t bs = scalar_num_bits(s), bn = scalar_num_bits(&neg);
    uint8_t g;
    if (bs <= 1) g = 0; else if (bn <= 1) g = 1; else if (bs <= 8) g = 2; else if (bn <= 8) g = 3;
    else if (bs <= 16) g = 4; else if (bn <= 16) g = 5; else if (bs <= 32) g = 6; else if (bn <= 32) g = 7;
    else if (bs <= 64) g = 8; else if (bn <= 64) g = 9; else g = 10;
    group[i] = g;
  }
  for (size_t i = 0; i < n; i++) if (group[i] != 255) counts[group[i]]++;
  /* Phase 2: partition by group (:285-301; the sort key is the group only) */
  size_t start[12]; start[0] = 0;
  for (int g = 0; g < 11; g++) start[g + 1] = start[g] + counts[g];
  size_t m = start[11];
  if (m == 0) { free(group); return; }                              /* :281-283 */
  aff* gb = (aff*)malloc(sizeof(aff) * m);
  uint64_t* gs = (uint64_t*)malloc(sizeof(uint64_t) * m);
  fe* gl = (fe*)malloc(sizeof(fe) * (counts[10] ? counts[10] : 1));
  size_t pos[12]; memcpy(pos, start, sizeof(pos));
  for (size_t i = 0; i < n; i++) {
    uint8_t g = group[i];
    if (g == 255) continue;
    size_t o = pos[g]++;
    gb[o] = bases[i];
    if (g == 10) { gl[o - start[10]] = coeffs[i]; gs[o] = 0; }
    else if (g & 1) { fe neg; uint64_t bw; fe_sub_raw(&neg, &S->p, &coeffs[i], &bw); gs[o] = neg.l[0]; } /* repr_low_u64(-s) */
    else gs[o] = coeffs[i].l[0];
  }
  /* Phase 3 (:324-418) */
  xyzz pos_r, neg_r, total; xyzz_zero(F, &total);
  accumulate_bases(C, gb + start[0], counts[0], &pos_r);
  accumulate_bases(C, gb + start[1], counts[1], &neg_r);
  xyzz_sub(F, &pos_r, &neg_r); xyzz_add(F, &total, &pos_r);
  static const size_t BITS[4] = {8, 16, 32, 64};
  for (int k = 0; k < 4; k++) {
    int gp = 2 + 2 * k, gn = gp + 1;
    msm_small_bits(C, gs + start[gp], gb + start[gp], counts[gp], BITS[k], &pos_r);
    msm_small_bits(C, gs + start[gn], gb + start[gn], counts[gn], BITS[k], &neg_r);
    xyzz_sub(F, &pos_r, &neg_r); xyzz_add(F, &total, &pos_r);
  }
  if (counts[10]) { best_msm(C, gl, gb + start[10], counts[10], &pos_r); xyzz_add(F, &total, &pos_r); }
  free(group); free(gb); free(gs); free(gl);
  *out = total;
}

/* ------------------------------------------------------------------ marshalling + exported API ---------- */
static void load_bases(const curve_t* C, const uint8_t* xy64, size_t n, aff* out) {
#pragma omp parallel for num_threads(nthreads())
  for (long i = 0; i < (long)n; i++) {
    fe x, y;
    memcpy(&x, xy64 + 64 * i, 32); memcpy(&y, xy64 + 64 * i + 32, 32);
    fe_to_mont(C->base, &out[i].x, &x); fe_to_mont(C->base, &out[i].y, &y);
  }
}
static void store_point(const curve_t* C, const xyzz* p, uint8_t* out, uint8_t* is_inf) {
  aff a; xyzz_to_affine(C->base, &a, p);
  if (xyzz_is_zero(p)) { memset(out, 0, 64); if (is_inf) *is_inf = 1; return; }
  fe x, y; fe_from_mont(C->base, &x, &a.x); fe_from_mont(C->base, &y, &a.y);
  memcpy(out, &x, 32); memcpy(out + 32, &y, 32);
  if (is_inf) *is_inf = 0;
}

void ref_set_threads(int t) { g_threads = t; }
int ref_get_threads(void) { return nthreads(); }

/* Opaque prepared key: bases converted once to Montgomery form (a host `Vec<Affine>` already is) */
void* ref_bases_load(int curve, const uint8_t* bases_xy64, size_t n) {
  if (curve < 0 || curve > 3) return NULL;
  aff* b = (aff*)malloc(sizeof(aff) * (n ? n : 1));
  load_bases(&CURVES[curve], bases_xy64, n, b);
  return b;
}
void ref_bases_free(void* h) { free(h); }

/* msm() on a prepared key prefix; scalars canonical LE 32 bytes */
int ref_msm_prepared(int curve, const uint8_t* scalars_le32, const void* prepared, size_t n, uint8_t* out, uint8_t* is_inf) {
  if (curve < 0 || curve > 3) return -1;
  const curve_t* C = &CURVES[curve];
  for (size_t i = 0; i < n; i++) { /* from_repr would reject s >= r */
    fe s; memcpy(&s, scalars_le32 + 32 * i, 32);
    if (fe_geq(&s, &C->scalar->p)) return -4;
  }
  xyzz r; msm_full(C, (const fe*)scalars_le32, (const aff*)prepared, n, &r);
  store_point(C, &r, out, is_inf);
  return 0;
}
int ref_msm(int curve, const uint8_t* scalars_le32, const uint8_t* bases_xy64, size_t n, uint8_t* out, uint8_t* is_inf) {
  void* b = ref_bases_load(curve, bases_xy64, n);
  if (!b) return -1;
  int rc = ref_msm_prepared(curve, scalars_le32, b, n, out, is_inf);
  free(b);
  return rc;
}
/* halo2curves::msm::msm_best role only (benches/commit.rs:115 times this directly) */
int ref_msm_best_prepared(int curve, const uint8_t* scalars_le32, const void* prepared, size_t n, uint8_t* out, uint8_t* is_inf) {
  if (curve < 0 || curve > 3) return -1;
  const curve_t* C = &CURVES[curve];
  xyzz r; best_msm(C, (const fe*)scalars_le32, (const aff*)prepared, n, &r);
  store_point(C, &r, out, is_inf);
  return 0;
}
/* msm_small_with_max_num_bits; max_num_bits == (size_t)-1 -> msm_small (msm.rs:469-475) */
int ref_msm_u64_prepared(int curve, const uint64_t* s, const void* prepared, size_t n, size_t max_num_bits, uint8_t* out, uint8_t* is_inf) {
  if (curve < 0 || curve > 3) return -1;
  const curve_t* C = &CURVES[curve];
  if (max_num_bits == (size_t)-1) {
    uint64_t mx = 0; for (size_t i = 0; i < n; i++) if (s[i] > mx) mx = s[i];
    max_num_bits = num_bits_usize(mx);
  }
  if (max_num_bits >= 2 && max_num_bits <= 10)
    for (size_t i = 0; i < n; i++) if (s[i] >> max_num_bits) return -5; /* msm.rs:552 would index out of bounds */
  xyzz r; msm_small_bits(C, s, (const aff*)prepared, n, max_num_bits, &r);
  store_point(C, &r, out, is_inf);
  return 0;
}
int ref_msm_u64(int curve, const uint64_t* s, const uint8_t* bases_xy64, size_t n, size_t max_num_bits, uint8_t* out, uint8_t* is_inf) {
  void* b = ref_bases_load(curve, bases_xy64, n);
  if (!b) return -1;
  int rc = ref_msm_u64_prepared(curve, s, b, n, max_num_bits, out, is_inf);
  free(b);
  return rc;
}
/* batch_vartime_multiscalar_mul (traits.rs:82-90) */
int ref_msm_batch(int curve, const uint8_t* const* vecs, const size_t* lens, size_t k, const uint8_t* bases_xy64, size_t n_bases,
                  uint8_t* out, uint8_t* is_inf) {
  void* b = ref_bases_load(curve, bases_xy64, n_bases);
  if (!b) return -1;
  int rc = 0;
  for (size_t j = 0; j < k && rc == 0; j++) {
    if (lens[j] > n_bases) { rc = -1; break; }
    rc = ref_msm_prepared(curve, vecs[j], b, lens[j], out + 64 * j, is_inf ? is_inf + j : NULL);
  }
  free(b);
  return rc;
}
/* commit = msm(v, ck[..n]) + h * r (pedersen.rs:263-270) */
int ref_commit(int curve, const uint8_t* v_le32, const uint8_t* ck_xy64, size_t n, const uint8_t* h_xy64, const uint8_t* r_le32,
               uint8_t* out, uint8_t* is_inf) {
  if (curve < 0 || curve > 3) return -1;
  const curve_t* C = &CURVES[curve];
  aff* b = (aff*)ref_bases_load(curve, ck_xy64, n);
  xyzz acc; msm_full(C, (const fe*)v_le32, b, n, &acc);
  free(b);
  aff h; load_bases(C, h_xy64, 1, &h);
  fe r; memcpy(&r, r_le32, 32);
  xyzz t; xyzz_scalar_mul(C->base, &t, &h, &r);
  xyzz_add(C->base, &acc, &t);
  store_point(C, &acc, out, is_inf);
  return 0;
}
/* batch_add (msm.rs:689-708): sum of bases at the given indices */
int ref_batch_add(int curve, const uint8_t* bases_xy64, size_t n_bases, const uint64_t* idx, size_t k, uint8_t* out, uint8_t* is_inf) {
  if (curve < 0 || curve > 3) return -1;
  const curve_t* C = &CURVES[curve];
  aff* b = (aff*)ref_bases_load(curve, bases_xy64, n_bases);
  xyzz acc; xyzz_zero(C->base, &acc);
  for (size_t i = 0; i < k; i++) xyzz_add_affine(C->base, &acc, &b[idx[i]]);
  free(b);
  store_point(C, &acc, out, is_inf);
  return 0;
}
/* P_i = (k0 + i) * G (curve_property_tests.rs:186-194); generator given by the caller as canonical x||y */
int ref_sequential_bases(int curve, const uint8_t* gen_xy64, uint64_t k0, size_t n, uint8_t* out_xy64) {
  if (curve < 0 || curve > 3) return -1;
  const curve_t* C = &CURVES[curve];
  aff g; load_bases(C, gen_xy64, 1, &g);
  int nt = nthreads();
  size_t chunk = (n + nt - 1) / (size_t)nt;
#pragma omp parallel for num_threads(nt)
  for (int t = 0; t < nt; t++) {
    size_t lo = (size_t)t * chunk, hi = lo + chunk < n ? lo + chunk : n;
    if (lo >= hi) continue;
    fe k = {{k0 + lo, 0, 0, 0}};
    xyzz p; xyzz_scalar_mul(C->base, &p, &g, &k);
    for (size_t i = lo; i < hi; i++) {
      store_point(C, &p, out_xy64 + 64 * i, NULL);
      xyzz_add_affine(C->base, &p, &g);
    }
  }
  return 0;
}
/* field-vector kernels of the "next" rows, canonical in/out: out = a + r*b  (r1cs/mod.rs:1058-1067) */
int ref_field_axpy(int field, const uint8_t* a, const uint8_t* b, const uint8_t* r, size_t n, uint8_t* out) {
  static const field_t* FS[4] = {&F_BN_Q, &F_BN_R, &F_PA_P, &F_PA_Q};
  if (field < 0 || field > 3) return -1;
  const field_t* F = FS[field];
  fe rr; memcpy(&rr, r, 32); fe_to_mont(F, &rr, &rr);
#pragma omp parallel for num_threads(nthreads())
  for (long i = 0; i < (long)n; i++) {
    fe x, y; memcpy(&x, a + 32 * i, 32); memcpy(&y, b + 32 * i, 32);
    fe_mul(F, &y, &y, &rr); /* (y * rR)/R = y*r, canonical */
    fe_add(F, &x, &x, &y);
    memcpy(out + 32 * i, &x, 32);
  }
  return 0;
}

/* ---- remaining field-vector restatements, canonical in/out ---------------------------- */
static const field_t* field_by_id(int f) {
  static const field_t* FS[4] = {&F_BN_Q, &F_BN_R, &F_PA_P, &F_PA_Q};
  return (f < 0 || f > 3) ? NULL : FS[f];
}
static inline void ld_mont(const field_t* F, fe* r, const uint8_t* p) { fe t; memcpy(&t, p, 32); fe_to_mont(F, r, &t); }
static inline void st_canon(const field_t* F, uint8_t* p, const fe* v) { fe t; fe_from_mont(F, &t, v); memcpy(p, &t, 32); }
/* E = E1 + r*T + r^2*E2 (r1cs/mod.rs:1096-1101) */
int ref_field_axpy2(int field, const uint8_t* a, const uint8_t* b, const uint8_t* c, const uint8_t* r, size_t n, uint8_t* out) {
  const field_t* F = field_by_id(field); if (!F) return -1;
  fe rr, rr2; ld_mont(F, &rr, r); fe_mul(F, &rr2, &rr, &rr);
#pragma omp parallel for num_threads(nthreads())
  for (long i = 0; i < (long)n; i++) {
    fe x, y, z, t; ld_mont(F, &x, a + 32 * i); ld_mont(F, &y, b + 32 * i); ld_mont(F, &z, c + 32 * i);
    fe_mul(F, &t, &rr, &y); fe_add(F, &x, &x, &t); fe_mul(F, &t, &rr2, &z); fe_add(F, &x, &x, &t);
    st_canon(F, out + 32 * i, &x);
  }
  return 0;
}
/* T = az*bz - u*cz - e (r1cs/mod.rs:614-620) */
int ref_field_cross_term(int field, const uint8_t* az, const uint8_t* bz, const uint8_t* cz, const uint8_t* e, const uint8_t* u,
                         size_t n, uint8_t* out) {
  const field_t* F = field_by_id(field); if (!F) return -1;
  fe uu; ld_mont(F, &uu, u);
#pragma omp parallel for num_threads(nthreads())
  for (long i = 0; i < (long)n; i++) {
    fe a, b, c, ee, t; ld_mont(F, &a, az + 32 * i); ld_mont(F, &b, bz + 32 * i); ld_mont(F, &c, cz + 32 * i); ld_mont(F, &ee, e + 32 * i);
    fe_mul(F, &a, &a, &b); fe_mul(F, &t, &uu, &c); fe_sub(F, &a, &a, &t); fe_sub(F, &a, &a, &ee);
    st_canon(F, out + 32 * i, &a);
  }
  return 0;
}
/* T = az*bz - u*cz - e1 - e2 (commit_T_relaxed, r1cs/mod.rs:652-659) */
int ref_field_cross_term2(int field, const uint8_t* az, const uint8_t* bz, const uint8_t* cz, const uint8_t* e1, const uint8_t* e2,
                          const uint8_t* u, size_t n, uint8_t* out) {
  const field_t* F = field_by_id(field); if (!F) return -1;
  fe uu; ld_mont(F, &uu, u);
#pragma omp parallel for num_threads(nthreads())
  for (long i = 0; i < (long)n; i++) {
    fe a, b, c, x, y, t; ld_mont(F, &a, az + 32 * i); ld_mont(F, &b, bz + 32 * i); ld_mont(F, &c, cz + 32 * i);
    ld_mont(F, &x, e1 + 32 * i); ld_mont(F, &y, e2 + 32 * i);
    fe_mul(F, &a, &a, &b); fe_mul(F, &t, &uu, &c); fe_sub(F, &a, &a, &t); fe_sub(F, &a, &a, &x); fe_sub(F, &a, &a, &y);
    st_canon(F, out + 32 * i, &a);
  }
  return 0;
}
/* out[i] = z[lo + i*stride] + r*(z[hi + i*stride] - z[lo + i*stride]):
 * bind_poly_var_top (multilinear.rs:65-84: lo = 0, hi = len/2, stride 1) and the HyperKZG halving
 * (hyperkzg.rs:1085-1095: lo = 0, hi = 1, stride 2) */
int ref_field_bind(int field, const uint8_t* z, size_t lo, size_t hi, size_t stride, const uint8_t* r, size_t n_out, uint8_t* out) {
  const field_t* F = field_by_id(field); if (!F) return -1;
  fe rr; ld_mont(F, &rr, r);
#pragma omp parallel for num_threads(nthreads())
  for (long i = 0; i < (long)n_out; i++) {
    fe a, b, t; ld_mont(F, &a, z + 32 * (lo + i * stride)); ld_mont(F, &b, z + 32 * (hi + i * stride));
    fe_sub(F, &t, &b, &a); fe_mul(F, &t, &rr, &t); fe_add(F, &a, &a, &t);
    st_canon(F, out + 32 * i, &a);
  }
  return 0;
}

/* (t_0, t_inf) of the eq-factored sum-check rounds (src/spartan/sumcheck.rs:900-958 mode 3, :972-1037 mode 2,
 * :1039-1075 mode 1); eqL == NULL selects the last-half form (factor = eqR[id]).  Canonical in/out. */
int ref_sumcheck_eq_sums(int field, int mode, const uint8_t* A, const uint8_t* B, const uint8_t* C, size_t len, const uint8_t* eqL,
                         const uint8_t* eqR, unsigned shift, uint8_t* out64) {
  const field_t* F = field_by_id(field); if (!F) return -1;
  size_t h = len / 2, mask = ((size_t)1 << shift) - 1;
  fe s0, s1; memset(&s0, 0, sizeof s0); memset(&s1, 0, sizeof s1);
  for (size_t id = 0; id < h; id++) {
    fe fac, t; ld_mont(F, &fac, eqR + 32 * (eqL ? (id & mask) : id));
    if (eqL) { ld_mont(F, &t, eqL + 32 * (id >> shift)); fe_mul(F, &fac, &t, &fac); }
    fe a0; ld_mont(F, &a0, A + 32 * id);
    if (mode == 1) { fe_mul(F, &t, &a0, &fac); fe_add(F, &s0, &s0, &t); continue; }
    fe a1, b0, b1, e0, q, c;
    ld_mont(F, &a1, A + 32 * (id + h)); ld_mont(F, &b0, B + 32 * id); ld_mont(F, &b1, B + 32 * (id + h));
    if (mode == 3) ld_mont(F, &c, C + 32 * id); else c = F->r1;
    fe_mul(F, &e0, &a0, &b0); fe_sub(F, &e0, &e0, &c);
    fe_sub(F, &a1, &a1, &a0); fe_sub(F, &b1, &b1, &b0); fe_mul(F, &q, &a1, &b1);
    fe_mul(F, &t, &e0, &fac); fe_add(F, &s0, &s0, &t);
    fe_mul(F, &t, &q, &fac); fe_add(F, &s1, &s1, &t);
  }
  st_canon(F, out64, &s0); st_canon(F, out64 + 32, &s1);
  return 0;
}

/* EqPolynomial::evals_from_points (src/spartan/polys/eq.rs:54-73), canonical in/out */
int ref_eq_evals(int field, const uint8_t* r, size_t ell, uint8_t* out) {
  const field_t* F = field_by_id(field); if (!F) return -1;
  size_t n = (size_t)1 << ell;
  fe* ev = (fe*)calloc(n, sizeof(fe));
  ev[0] = F->r1;
  size_t size = 1;
  for (size_t j = ell; j-- > 0;) {           /* for r in r.iter().rev() */
    fe rr; ld_mont(F, &rr, r + 32 * j);
#pragma omp parallel for num_threads(nthreads()) if (size >= 4096)   /* zip_with_for_each!(par_iter_mut, ..), eq.rs:64-67 */
    for (long i = 0; i < (long)size; i++) {
      fe y; fe_mul(F, &y, &ev[i], &rr);       /* *y = *x * r */
      ev[i + size] = y;
      fe_sub(F, &ev[i], &ev[i], &y);          /* *x -= *y */
    }
    size *= 2;
  }
#pragma omp parallel for num_threads(nthreads()) if (n >= 4096)
  for (long i = 0; i < (long)n; i++) st_canon(F, out + 32 * i, &ev[i]);
  free(ev);
  return 0;
}
/* MultilinearPolynomial::evaluate_with (src/spartan/polys/multilinear.rs:98-129): sqrt decomposition */
int ref_mle_evaluate(int field, const uint8_t* z, size_t ell, const uint8_t* r, uint8_t* out32) {
  const field_t* F = field_by_id(field); if (!F) return -1;
  size_t s_right = ell / 2, s_left = ell - s_right, n_left = (size_t)1 << s_left, n_right = (size_t)1 << s_right;
  uint8_t* el = (uint8_t*)malloc(32 * n_left); uint8_t* er = (uint8_t*)malloc(32 * n_right);
  ref_eq_evals(field, r, s_left, el); ref_eq_evals(field, r + 32 * s_left, s_right, er);
  fe acc; memset(&acc, 0, sizeof acc);
  fe* terms = (fe*)malloc(n_left * sizeof(fe));
#pragma omp parallel for num_threads(nthreads()) if (n_left * n_right >= 4096)   /* (0..n_left).into_par_iter(), multilinear.rs:110-120 */
  for (long i = 0; i < (long)n_left; i++) {
    fe red; memset(&red, 0, sizeof red);
    for (size_t j = 0; j < n_right; j++) {
      fe zz, e, t; ld_mont(F, &zz, z + 32 * ((size_t)i * n_right + j)); ld_mont(F, &e, er + 32 * j);
      fe_mul(F, &t, &zz, &e); fe_add(F, &red, &red, &t);
    }
    fe e; ld_mont(F, &e, el + 32 * i); fe_mul(F, &terms[i], &e, &red);
  }
  for (size_t i = 0; i < n_left; i++) fe_add(F, &acc, &acc, &terms[i]);   /* (field addition is exact: any order gives the same element) */
  st_canon(F, out32, &acc);
  free(el); free(er); free(terms);
  return 0;
}
/* SparseMatrix::multiply_vec (src/r1cs/sparse.rs:201-229), CSR with usize indices */
int ref_spmv(int field, const uint64_t* indptr, const uint64_t* indices, const uint8_t* data, size_t rows, const uint8_t* z, uint8_t* out) {
  const field_t* F = field_by_id(field); if (!F) return -1;
#pragma omp parallel for num_threads(nthreads())
  for (long rw = 0; rw < (long)rows; rw++) {
    fe acc; memset(&acc, 0, sizeof acc);
    for (uint64_t k = indptr[rw]; k < indptr[rw + 1]; k++) {
      fe d, v, t; ld_mont(F, &d, data + 32 * k); ld_mont(F, &v, z + 32 * indices[k]);
      fe_mul(F, &t, &d, &v); fe_add(F, &acc, &acc, &t);
    }
    st_canon(F, out + 32 * rw, &acc);
  }
  return 0;
}

/* PrecomputedSparseMatrix::multiply_vec_pair (src/r1cs/sparse.rs:215-229): (M*v1, M*v2), one pass over the rows */
int ref_spmv_pair(int field, const uint64_t* indptr, const uint64_t* indices, const uint8_t* data, size_t rows,
                  const uint8_t* z1, const uint8_t* z2, uint8_t* out1, uint8_t* out2) {
  const field_t* F = field_by_id(field); if (!F) return -1;
#pragma omp parallel for num_threads(nthreads())
  for (long rw = 0; rw < (long)rows; rw++) {
    fe a1, a2; memset(&a1, 0, sizeof a1); memset(&a2, 0, sizeof a2);
    for (uint64_t k = indptr[rw]; k < indptr[rw + 1]; k++) {
      fe d, v, t; ld_mont(F, &d, data + 32 * k);
      ld_mont(F, &v, z1 + 32 * indices[k]); fe_mul(F, &t, &d, &v); fe_add(F, &a1, &a1, &t);
      ld_mont(F, &v, z2 + 32 * indices[k]); fe_mul(F, &t, &d, &v); fe_add(F, &a2, &a2, &t);
    }
    st_canon(F, out1 + 32 * rw, &a1); st_canon(F, out2 + 32 * rw, &a2);
  }
  return 0;
}

/* The sum-check round sums without an eq factor (src/spartan/sumcheck.rs): kind 1 compute_eval_points_quad_prod
 * (:163-186), 2 compute_eval_points_linear (:353-378), 3 compute_eval_points_quadratic (:380-405),
 * 4 compute_eval_points_cubic (:407-443).  out96 = three canonical elements (third zero for kinds 1-3). */
int ref_sumcheck_plain_sums(int field, int kind, const uint8_t* A, const uint8_t* B, const uint8_t* C, size_t len, uint8_t* out96) {
  const field_t* F = field_by_id(field); if (!F) return -1;
  size_t h = len / 2;
  fe s0, s1, s2; memset(&s0, 0, sizeof s0); memset(&s1, 0, sizeof s1); memset(&s2, 0, sizeof s2);
  for (size_t i = 0; i < h; i++) {
    fe a0, a1, b0, b1, c0, c1, t, u, dA, dB, dC, am, bm, cm;
    ld_mont(F, &a0, A + 32 * i); ld_mont(F, &a1, A + 32 * (i + h));
    ld_mont(F, &b0, B + 32 * i); ld_mont(F, &b1, B + 32 * (i + h));
    fe_sub(F, &dA, &a1, &a0); fe_sub(F, &dB, &b1, &b0);
    fe_add(F, &am, &a0, &a0); fe_sub(F, &am, &am, &a1);      /* A(-1) = a0 + a0 - a1 */
    fe_add(F, &bm, &b0, &b0); fe_sub(F, &bm, &bm, &b1);
    switch (kind) {
      case 1: fe_mul(F, &t, &a0, &b0); fe_add(F, &s0, &s0, &t); fe_mul(F, &t, &dA, &dB); fe_add(F, &s1, &s1, &t); break;
      case 2: fe_sub(F, &t, &a0, &b0); fe_add(F, &s0, &s0, &t); fe_sub(F, &t, &am, &bm); fe_add(F, &s1, &s1, &t); break;
      case 3: fe_mul(F, &t, &a0, &b0); fe_add(F, &s0, &s0, &t); fe_mul(F, &t, &am, &bm); fe_add(F, &s1, &s1, &t); break;
      case 4:
        ld_mont(F, &c0, C + 32 * i); ld_mont(F, &c1, C + 32 * (i + h));
        fe_sub(F, &dC, &c1, &c0);
        fe_sub(F, &am, &a0, &dA); fe_sub(F, &bm, &b0, &dB); fe_sub(F, &cm, &c0, &dC);   /* (poly[i] - d), :432 */
        fe_mul(F, &t, &a0, &b0); fe_mul(F, &t, &t, &c0); fe_add(F, &s0, &s0, &t);
        fe_mul(F, &u, &dA, &dB); fe_mul(F, &u, &u, &dC); fe_add(F, &s1, &s1, &u);
        fe_mul(F, &t, &am, &bm); fe_mul(F, &t, &t, &cm); fe_add(F, &s2, &s2, &t);
        break;
      default: return -1;
    }
  }
  st_canon(F, out96, &s0); st_canon(F, out96 + 32, &s1); st_canon(F, out96 + 64, &s2);
  return 0;
}

/* PolyEvalWitness::batch / batch_diff_size (src/spartan/mod.rs:165-277): out[i] = sum_j s^j * vecs[j][i], shorter
 * vectors zero-padded; powers::<E>(s, k) = [1, s, s^2, ...] (:130-137) */
int ref_lincomb_powers(int field, const uint8_t* const* vecs, const size_t* lens, size_t k, const uint8_t* s, size_t n_out, uint8_t* out) {
  const field_t* F = field_by_id(field); if (!F) return -1;
  fe sm; ld_mont(F, &sm, s);
  fe* acc = (fe*)calloc(n_out ? n_out : 1, sizeof(fe));
  fe pw = F->r1;
  for (size_t j = 0; j < k; j++) {
    const size_t lj = lens[j];
#pragma omp parallel for num_threads(nthreads()) if (lj >= 4096)   /* par chunks / par_iter_mut, spartan/mod.rs:170-277 */
    for (long i = 0; i < (long)lj; i++) {
      fe v, t; ld_mont(F, &v, vecs[j] + 32 * i);
      fe_mul(F, &t, &pw, &v); fe_add(F, &acc[i], &acc[i], &t);
    }
    fe_mul(F, &pw, &pw, &sm);
  }
#pragma omp parallel for num_threads(nthreads()) if (n_out >= 4096)
  for (long i = 0; i < (long)n_out; i++) st_canon(F, out + 32 * i, &acc[i]);
  free(acc);
  return 0;
}

/* MultilinearPolynomial::multi_evaluate_with (src/spartan/polys/multilinear.rs:131-180): row sums of all k
 * polynomials against eq_right, then a dot with eq_left */
int ref_mle_multi_evaluate(int field, const uint8_t* const* zs, size_t k, size_t ell, const uint8_t* r, uint8_t* out) {
  const field_t* F = field_by_id(field); if (!F) return -1;
  if (k == 0) return 0;
  size_t s_right = ell / 2, s_left = ell - s_right, n_left = (size_t)1 << s_left, n_right = (size_t)1 << s_right;
  uint8_t* el = (uint8_t*)malloc(32 * n_left); uint8_t* er = (uint8_t*)malloc(32 * n_right);
  ref_eq_evals(field, r, s_left, el); ref_eq_evals(field, r + 32 * s_left, s_right, er);
  fe* red = (fe*)calloc(n_left * k, sizeof(fe));
#pragma omp parallel for num_threads(nthreads()) if (n_left * n_right >= 4096)   /* all_reduced.par_chunks_mut(k), multilinear.rs:152-163 */
  for (long i = 0; i < (long)n_left; i++)
    for (size_t j = 0; j < n_right; j++) {
      fe e; ld_mont(F, &e, er + 32 * j);
      for (size_t p = 0; p < k; p++) {
        fe zz, t; ld_mont(F, &zz, zs[p] + 32 * ((size_t)i * n_right + j));
        fe_mul(F, &t, &zz, &e); fe_add(F, &red[i * k + p], &red[i * k + p], &t);
      }
    }
  for (size_t p = 0; p < k; p++) {
    fe acc; memset(&acc, 0, sizeof acc);
    for (size_t i = 0; i < n_left; i++) {
      fe e, t; ld_mont(F, &e, el + 32 * i); fe_mul(F, &t, &e, &red[i * k + p]); fe_add(F, &acc, &acc, &t);
    }
    st_canon(F, out + 32 * p, &acc);
  }
  free(el); free(er); free(red);
  return 0;
}

/* div_by_monomial (src/provider/hyperkzg.rs:946-999: h[i-1] = f[i] + h[i]*u, chunked in the reference, serial here)
 * and Horner poly_eval (hyperkzg.rs:1011-1020) in one pass: out[i] = sum_{k>=i} f[k] u^(k-i). */
int ref_poly_suffix_horner(int field, const uint8_t* f, size_t n, const uint8_t* u, uint8_t* out) {
  const field_t* F = field_by_id(field); if (!F) return -1;
  fe uu, acc; ld_mont(F, &uu, u); memset(&acc, 0, sizeof acc);
  const size_t T = (size_t)nthreads();
  if (n < 8192 || T < 2) {
    for (size_t i = n; i-- > 0;) {
      fe fi; ld_mont(F, &fi, f + 32 * i);
      fe_mul(F, &acc, &acc, &uu); fe_add(F, &acc, &acc, &fi);   /* acc = acc * u + fi */
      st_canon(F, out + 32 * i, &acc);
    }
    return 0;
  }
  /* Chunked as the reference's div_by_monomial is (hyperkzg.rs:961-999: per-chunk Horner, chunk heads chained with u^chunk, a second
   * pass adds the carry): chunk c = [c L, min(n, (c + 1) L)); local[i] = sum_{k >= i, k in chunk} f[k] u^(k - i); the value entering
   * chunk c from the right is carry[c] = out[(c + 1) L]; out[i] = local[i] + carry[c] u^(end_c - i).  Same elements as the serial pass. */
  const size_t L = (n + T - 1) / T, C = (n + L - 1) / L;
  fe* loc = (fe*)malloc(n * sizeof(fe));
  fe* carry = (fe*)calloc(C + 1, sizeof(fe));
#pragma omp parallel for num_threads(nthreads()) schedule(static, 1)
  for (long c = 0; c < (long)C; c++) {
    const size_t lo = (size_t)c * L, hi = lo + L < n ? lo + L : n;
    fe a; memset(&a, 0, sizeof a);
    for (size_t i = hi; i-- > lo;) {
      fe fi; ld_mont(F, &fi, f + 32 * i);
      fe_mul(F, &a, &a, &uu); fe_add(F, &a, &a, &fi);
      loc[i] = a;
    }
  }
  for (size_t c = C; c-- > 0;) {   /* carry[c] = out[(c + 1) L] = local head of chunk c + 1 + carry[c + 1] u^len(c + 1) */
    if (c + 1 >= C) continue;      /* (the last chunk has nothing to its right: carry stays zero) */
    const size_t lo = (c + 1) * L, hi = lo + L < n ? lo + L : n;
    fe pw = F->r1, t;
    for (size_t i = lo; i < hi; i++) fe_mul(F, &pw, &pw, &uu);
    fe_mul(F, &t, &carry[c + 1], &pw); fe_add(F, &carry[c], &loc[lo], &t);
  }
#pragma omp parallel for num_threads(nthreads()) schedule(static, 1)
  for (long c = 0; c < (long)C; c++) {
    const size_t lo = (size_t)c * L, hi = lo + L < n ? lo + L : n;
    fe pw = carry[c];                          /* carry u^(hi - i), built from the right end of the chunk */
    for (size_t i = hi; i-- > lo;) {
      fe v; fe_mul(F, &pw, &pw, &uu); fe_add(F, &v, &loc[i], &pw);
      st_canon(F, out + 32 * i, &v);
    }
  }
  free(loc); free(carry);
  return 0;
}

/* ======================================================================================================================
 * Spartan's sum-check provers and compute_eval_table_sparse (round 5: BASELINE.json configs[4], the sum-check half).
 * Restated from /root/reference/src/spartan/sumcheck.rs and src/spartan/mod.rs; the transcript (Keccak, src/provider/keccak.rs)
 * is NOT restated: every prover takes a callback that plays `transcript.absorb(b"p", &poly); transcript.squeeze(b"c")`
 * (sumcheck.rs:224-227,481-484,315-318) -- it receives the round polynomial's coefficients (UniPoly, little endian, canonical
 * 32-byte elements) and returns the round challenge.  All vectors canonical in / out.  OpenMP over the N-scaling sums
 * (the reference uses rayon there).
 * ====================================================================================================================== */
typedef int (*ref_transcript_fn)(void* ctx, const uint8_t* coeffs, size_t n_coeffs, uint8_t* challenge32);

/* compute_eval_table_sparse's inner (src/spartan/mod.rs:506-512): M_evals[col] += rx[row] * val, i.e. out = M^T rx.
 * rows = rx.len() = num_cons, out has `cols` entries (2 * num_vars in the caller). */
int ref_spmv_transposed(int field, const uint64_t* indptr, const uint64_t* indices, const uint8_t* data, size_t rows, size_t cols,
                        const uint8_t* rx, uint8_t* out) {
  const field_t* F = field_by_id(field); if (!F) return -1;
  const size_t nnz = (size_t)indptr[rows];
  if (nnz < 8192 || nthreads() < 2) {   /* the reference's loop as written (one matrix on one thread) */
    fe* acc = (fe*)calloc(cols ? cols : 1, sizeof(fe));
    for (size_t rw = 0; rw < rows; rw++) {
      fe x; ld_mont(F, &x, rx + 32 * rw);
      for (uint64_t k = indptr[rw]; k < indptr[rw + 1]; k++) {
        fe d, t; ld_mont(F, &d, data + 32 * k);
        fe_mul(F, &t, &x, &d); fe_add(F, &acc[indices[k]], &acc[indices[k]], &t);
      }
    }
    for (size_t i = 0; i < cols; i++) st_canon(F, out + 32 * i, &acc[i]);
    free(acc);
    return 0;
  }
  /* The reference runs this loop on one thread per matrix, the three matrices side by side (rayon::join, spartan/mod.rs:513-531).
   * The oracle is called once per matrix, so to stay a fair baseline it spreads ONE matrix over the cores instead: entries grouped
   * by column (a counting sort over the indices: integers only, one thread), then the columns in parallel.  Same sums (field
   * addition is exact in any order). */
  uint32_t* start = (uint32_t*)calloc(cols + 2, sizeof(uint32_t));
  for (size_t k = 0; k < nnz; k++) start[indices[k] + 2]++;
  for (size_t c = 0; c < cols; c++) start[c + 2] += start[c + 1];
  uint32_t* ent = (uint32_t*)malloc((nnz ? nnz : 1) * sizeof(uint32_t));  /* entry ids grouped by column */
  uint32_t* row_of = (uint32_t*)malloc((nnz ? nnz : 1) * sizeof(uint32_t));
  for (size_t rw = 0; rw < rows; rw++)
    for (uint64_t k = indptr[rw]; k < indptr[rw + 1]; k++) {
      const uint32_t q = start[indices[k] + 1]++;
      ent[q] = (uint32_t)k; row_of[q] = (uint32_t)rw;
    }
#pragma omp parallel for num_threads(nthreads()) schedule(dynamic, 1024)
  for (long c = 0; c < (long)cols; c++) {
    fe a; memset(&a, 0, sizeof a);
    for (uint32_t q = start[c]; q < start[c + 1]; q++) {
      fe x, d, t; ld_mont(F, &x, rx + 32 * (size_t)row_of[q]); ld_mont(F, &d, data + 32 * (size_t)ent[q]);
      fe_mul(F, &t, &x, &d); fe_add(F, &a, &a, &t);
    }
    st_canon(F, out + 32 * c, &a);
  }
  free(start); free(ent); free(row_of);
  return 0;
}

static fe* load_vec_mont(const field_t* F, const uint8_t* v, size_t n) {
  fe* o = (fe*)malloc((n ? n : 1) * sizeof(fe));
#pragma omp parallel for num_threads(nthreads())
  for (long i = 0; i < (long)n; i++) ld_mont(F, &o[i], v + 32 * i);
  return o;
}
/* MultilinearPolynomial::bind_poly_var_top (src/spartan/polys/multilinear.rs:65-84), in place on Montgomery values */
static void bind_top_fe(const field_t* F, fe* z, size_t len, const fe* r) {
  size_t n = len / 2;
#pragma omp parallel for num_threads(nthreads())
  for (long i = 0; i < (long)n; i++) {
    fe t; fe_sub(F, &t, &z[i + n], &z[i]); fe_mul(F, &t, r, &t); fe_add(F, &z[i], &z[i], &t);
  }
}
static void fe_from_u64(const field_t* F, fe* r, uint64_t v) { fe t; memset(&t, 0, sizeof t); t.l[0] = v; fe_to_mont(F, r, &t); }
static void fe_pow2(const field_t* F, fe* r, size_t e) { /* Scalar::from(2).pow_vartime([e]) */
  fe two; fe_from_u64(F, &two, 2); *r = F->r1;
  for (size_t i = 0; i < e; i++) fe_mul(F, r, r, &two);
}
/* UniPoly::evaluate (src/spartan/polys/univariate.rs:140-149) */
static void unipoly_eval(const field_t* F, fe* out, const fe* coeffs, size_t n, const fe* r) {
  fe eval = coeffs[0], power = *r, t;
  for (size_t i = 1; i < n; i++) { fe_mul(F, &t, &power, &coeffs[i]); fe_add(F, &eval, &eval, &t); fe_mul(F, &power, &power, r); }
  *out = eval;
}
static fe two_inv(const field_t* F) { fe two, r; fe_from_u64(F, &two, 2); fe_inv(F, &r, &two); return r; }
/* UniPoly::from_evals_deg2 (src/spartan/polys/univariate.rs:90-99): evals = [f(0), f(1), quadratic coefficient] -> [c, b, a] */
static void unipoly_from_evals_deg2(const field_t* F, const fe* e0, const fe* e1, const fe* a, fe co[3]) {
  co[0] = *e0; co[2] = *a;
  fe_sub(F, &co[1], e1, a); fe_sub(F, &co[1], &co[1], e0);       /* b = f(1) - a - c */
}
/* UniPoly::from_evals_deg3 (univariate.rs:103-113): evals = [f(0), f(1), cubic coefficient, f(-1)] -> [d, c, b, a] */
static void unipoly_from_evals_deg3(const field_t* F, const fe* e0, const fe* e1, const fe* a, const fe* em1, fe co[4]) {
  const fe tinv = two_inv(F);
  fe t;
  co[0] = *e0; co[3] = *a;
  fe_add(F, &t, e1, em1); fe_mul(F, &co[2], &t, &tinv); fe_sub(F, &co[2], &co[2], e0);          /* b = (f(1) + f(-1)) / 2 - d */
  fe_sub(F, &co[1], e1, a); fe_sub(F, &co[1], &co[1], e0); fe_sub(F, &co[1], &co[1], &co[2]);   /* c = f(1) - a - d - b */
}
/* test hook for the reference's known answers (univariate.rs:284-355: 2x^2+3x+1 from [1, 6, 2]; x^3+2x^2+3x+1 from [1, 7, 1, -1]):
 * evals = deg + 1 canonical elements in the order the constructors take them; coefficients (deg + 1) and the value at `at` out */
int ref_unipoly_from_evals(int field, int deg, const uint8_t* evals, const uint8_t* at, uint8_t* out_coeffs, uint8_t* out_value) {
  const field_t* F = field_by_id(field); if (!F || (deg != 2 && deg != 3)) return -1;
  fe* e = load_vec_mont(F, evals, (size_t)deg + 1);
  fe* x = load_vec_mont(F, at, 1);
  fe co[4], v;
  if (deg == 2) unipoly_from_evals_deg2(F, &e[0], &e[1], &e[2], co);
  else unipoly_from_evals_deg3(F, &e[0], &e[1], &e[2], &e[3], co);
  unipoly_eval(F, &v, co, (size_t)deg + 1, &x[0]);
  for (int i = 0; i <= deg; i++) st_canon(F, out_coeffs + 32 * i, &co[i]);
  st_canon(F, out_value, &v);
  free(e); free(x);
  return 0;
}
/* transcript round trip: coefficients out (canonical), challenge in */
static int ask_challenge(const field_t* F, ref_transcript_fn cb, void* ctx, const fe* coeffs, size_t n, fe* r, uint8_t* polys_out,
                         uint8_t* r_out) {
  uint8_t buf[4 * 32], ch[32];
  for (size_t i = 0; i < n; i++) st_canon(F, buf + 32 * i, &coeffs[i]);
  if (polys_out) memcpy(polys_out, buf, 32 * n);
  if (cb(ctx, buf, n, ch) != 0) return -1;
  if (r_out) memcpy(r_out, ch, 32);
  ld_mont(F, r, ch);
  return 0;
}

/* EqSumCheckInstance (src/spartan/sumcheck.rs:593-677) */
typedef struct {
  size_t init_num_vars, first_half, second_half, round;
  fe* taus;
  fe eval_eq_left;
  fe** poly_eq_left;  size_t n_left;   /* poly_eq_left[i] has 2^i entries, i in [0, n_left) */
  fe** poly_eq_right; size_t n_right;
  fe *eq0, *eq_slope, *eq_m1;          /* eq_tau_0_a_inf */
} eq_inst;
/* compute_eq_polynomials (sumcheck.rs:612-632): result[i + 1] = [prev - prev * tau_i, prev * tau_i] */
static fe** compute_eq_polynomials(const field_t* F, const fe* const* taus, size_t len) {
  fe** res = (fe**)malloc((len + 1) * sizeof(fe*));
  res[0] = (fe*)malloc(sizeof(fe)); res[0][0] = F->r1;
  for (size_t i = 0; i < len; i++) {
    size_t pl = (size_t)1 << i;
    res[i + 1] = (fe*)malloc(2 * pl * sizeof(fe));
    for (size_t j = 0; j < pl; j++) {
      fe y; fe_mul(F, &y, &res[i][j], taus[i]);
      res[i + 1][pl + j] = y;
      fe_sub(F, &res[i + 1][j], &res[i][j], &y);
    }
  }
  return res;
}
static eq_inst* eq_new(const field_t* F, const uint8_t* taus_le, size_t l) {
  eq_inst* q = (eq_inst*)calloc(1, sizeof(eq_inst));
  q->init_num_vars = l; q->first_half = l / 2; q->second_half = l - q->first_half; q->round = 1;
  q->taus = load_vec_mont(F, taus_le, l);
  q->eval_eq_left = F->r1;
  /* left_taus = taus[..first_half].iter().skip(1).rev(); right_taus = taus[first_half..].iter().rev()  (:634-636) */
  size_t nl = q->first_half > 0 ? q->first_half - 1 : 0, nr = q->second_half;
  const fe** lt = (const fe**)malloc((nl + 1) * sizeof(fe*));
  const fe** rt = (const fe**)malloc((nr + 1) * sizeof(fe*));
  for (size_t i = 0; i < nl; i++) lt[i] = &q->taus[q->first_half - 1 - i];
  for (size_t i = 0; i < nr; i++) rt[i] = &q->taus[l - 1 - i];
  q->poly_eq_left = compute_eq_polynomials(F, lt, nl); q->n_left = nl + 1;
  q->poly_eq_right = compute_eq_polynomials(F, rt, nr); q->n_right = nr + 1;
  free(lt); free(rt);
  q->eq0 = (fe*)malloc((l ? l : 1) * sizeof(fe)); q->eq_slope = (fe*)malloc((l ? l : 1) * sizeof(fe)); q->eq_m1 = (fe*)malloc((l ? l : 1) * sizeof(fe));
  for (size_t i = 0; i < l; i++) {   /* (:643-652) eq(tau, 0), 2 tau - 1, eq(tau, -1) */
    fe_sub(F, &q->eq0[i], &F->r1, &q->taus[i]);
    fe_sub(F, &q->eq_slope[i], &q->taus[i], &q->eq0[i]);
    fe_sub(F, &q->eq_m1[i], &q->eq0[i], &q->eq_slope[i]);
  }
  return q;
}
static void eq_free(eq_inst* q) {
  for (size_t i = 0; i < q->n_left; i++) free(q->poly_eq_left[i]);
  for (size_t i = 0; i < q->n_right; i++) free(q->poly_eq_right[i]);
  free(q->poly_eq_left); free(q->poly_eq_right); free(q->taus); free(q->eq0); free(q->eq_slope); free(q->eq_m1); free(q);
}
static void eq_bound(const field_t* F, eq_inst* q, const fe* r) { /* (:1226-1231) */
  fe t, u; const fe* tau = &q->taus[q->round - 1];
  fe_sub(F, &t, &F->r1, tau); fe_sub(F, &t, &t, r); fe_mul(F, &u, r, tau); fe_dbl(F, &u, &u); fe_add(F, &t, &t, &u);
  fe_mul(F, &q->eval_eq_left, &q->eval_eq_left, &t);
  q->round++;
}
/* the eq factor of index id in the current round (:1233-1253) */
static inline void eq_factor(const field_t* F, const eq_inst* q, size_t id, fe* fac) {
  if (q->round < q->first_half) {
    const fe* L = q->poly_eq_left[q->first_half - q->round];
    const fe* R = q->poly_eq_right[q->second_half];
    fe_mul(F, fac, &L[id >> q->second_half], &R[id & (((size_t)1 << q->second_half) - 1)]);
  } else {
    *fac = q->poly_eq_right[q->init_num_vars - q->round][id];
  }
}
/* N-scaling sums of one round: which = 0: (t_0, t_inf) of evaluation_points_* (:918-958, :1052-1075); which = 1: t(-1), the
 * third sum of the fallback_eval_inf_* paths (:1104-1130, :1204-1222).  mode 3: A*B - C; mode 1: A alone. */
static void eq_round_sums(const field_t* F, const eq_inst* q, int mode, int which, const fe* A, const fe* B, const fe* C, size_t len,
                          fe* o0, fe* o1) {
  size_t h = len / 2;
  int T = nthreads();
  fe* part = (fe*)calloc(2 * (size_t)T, sizeof(fe));
#pragma omp parallel num_threads(T)
  {
#ifdef _OPENMP
    int t = omp_get_thread_num(), nt = omp_get_num_threads();
#else
    int t = 0, nt = 1;
#endif
    fe s0, s1; memset(&s0, 0, sizeof s0); memset(&s1, 0, sizeof s1);
    for (size_t id = h * t / nt; id < h * (t + 1) / nt; id++) {
      fe fac, x, y, e;
      eq_factor(F, q, id, &fac);
      if (which == 0) {
        if (mode == 1) { fe_mul(F, &x, &A[id], &fac); fe_add(F, &s0, &s0, &x); continue; }
        fe_mul(F, &e, &A[id], &B[id]); fe_sub(F, &e, &e, &C[id]);
        fe_sub(F, &x, &A[id + h], &A[id]); fe_sub(F, &y, &B[id + h], &B[id]); fe_mul(F, &x, &x, &y);
        fe_mul(F, &e, &e, &fac); fe_add(F, &s0, &s0, &e);
        fe_mul(F, &x, &x, &fac); fe_add(F, &s1, &s1, &x);
      } else {
        fe ma, mb, mc;
        fe_dbl(F, &ma, &A[id]); fe_sub(F, &ma, &ma, &A[id + h]);
        if (mode == 1) { fe_mul(F, &x, &ma, &fac); fe_add(F, &s0, &s0, &x); continue; }
        fe_dbl(F, &mb, &B[id]); fe_sub(F, &mb, &mb, &B[id + h]);
        fe_dbl(F, &mc, &C[id]); fe_sub(F, &mc, &mc, &C[id + h]);
        fe_mul(F, &e, &ma, &mb); fe_sub(F, &e, &e, &mc); fe_mul(F, &e, &e, &fac); fe_add(F, &s0, &s0, &e);
      }
    }
    part[2 * t] = s0; part[2 * t + 1] = s1;
  }
  memset(o0, 0, sizeof *o0); if (o1) memset(o1, 0, sizeof *o1);
  for (int t = 0; t < T; t++) { fe_add(F, o0, o0, &part[2 * t]); if (o1) fe_add(F, o1, o1, &part[2 * t + 1]); }
  free(part);
}
/* evaluation_points_cubic_with_three_inputs (:900-970) / quadratic_with_one_input (:1039-1083): (s(0), cubic coeff, s(-1)) */
static void eq_eval_points(const field_t* F, const eq_inst* q, int mode, const fe* A, const fe* B, const fe* C, size_t len, const fe* claim,
                           fe* s0, fe* s_lead, fe* s_m1) {
  fe t0, tinf; memset(&tinf, 0, sizeof tinf);
  eq_round_sums(F, q, mode, 0, A, B, C, len, &t0, mode == 1 ? NULL : &tinf);
  const fe *eq0 = &q->eq0[q->round - 1], *slope = &q->eq_slope[q->round - 1], *eqm1 = &q->eq_m1[q->round - 1];
  const fe* p = &q->eval_eq_left;
  fe l0p, l1p, t;
  fe_mul(F, &l0p, eq0, p);
  fe_add(F, &t, eq0, slope); fe_mul(F, &l1p, &t, p);
  fe_mul(F, s0, &l0p, &t0);                                   /* s(0) = l(0) p t(0) */
  if (mode == 1) memset(s_lead, 0, sizeof *s_lead);
  else { fe_mul(F, s_lead, slope, p); fe_mul(F, s_lead, s_lead, &tinf); }
  fe tm1;
  if (!fe_is_zero(&l1p)) {                                     /* derive_from_claim_deg2 / _deg1 (:680-753) */
    fe inv, s1, t1; fe_inv(F, &inv, &l1p);
    fe_sub(F, &s1, claim, s0); fe_mul(F, &t1, &s1, &inv);
    fe_dbl(F, &tm1, &t0);
    if (mode != 1) { fe_dbl(F, &t, &tinf); fe_add(F, &tm1, &tm1, &t); }
    fe_sub(F, &tm1, &tm1, &t1);                                /* t(-1) = 2 t(inf) + 2 t(0) - t(1) */
  } else {                                                      /* tau = 0: the third N-scaling sum (:1085-1222) */
    eq_round_sums(F, q, mode, 1, A, B, C, len, &tm1, NULL);
  }
  fe_mul(F, s_m1, eqm1, p); fe_mul(F, s_m1, s_m1, &tm1);
}

/* SumcheckProof::prove_cubic_with_three_inputs (src/spartan/sumcheck.rs:446-507).  A, B, C: 2^num_rounds canonical elements each
 * (not modified: the reference binds its own copies in place).  out_polys: num_rounds x 4 coefficients (UniPoly, constant term
 * first); out_r: the challenges; out_claims: [A(r), B(r), C(r)]. */
int ref_sumcheck_prove_cubic3(int field, const uint8_t* claim, const uint8_t* taus, size_t num_rounds, const uint8_t* A, const uint8_t* B,
                              const uint8_t* C, ref_transcript_fn cb, void* ctx, uint8_t* out_polys, uint8_t* out_r, uint8_t* out_claims) {
  const field_t* F = field_by_id(field); if (!F) return -1;
  size_t len = (size_t)1 << num_rounds;
  fe *a = load_vec_mont(F, A, len), *b = load_vec_mont(F, B, len), *c = load_vec_mont(F, C, len);
  eq_inst* q = eq_new(F, taus, num_rounds);
  fe claim_per_round; ld_mont(F, &claim_per_round, claim);
  int rc = 0;
  for (size_t j = 0; j < num_rounds && rc == 0; j++) {
    fe s0, lead, sm1, s1;
    eq_eval_points(F, q, 3, a, b, c, len, &claim_per_round, &s0, &lead, &sm1);
    fe_sub(F, &s1, &claim_per_round, &s0);                     /* evals = [s0, claim - s0, lead, s(-1)] */
    fe co[4];
    unipoly_from_evals_deg3(F, &s0, &s1, &lead, &sm1, co);
    fe r;
    rc = ask_challenge(F, cb, ctx, co, 4, &r, out_polys ? out_polys + 128 * j : NULL, out_r ? out_r + 32 * j : NULL);
    if (rc) break;
    unipoly_eval(F, &claim_per_round, co, 4, &r);
    bind_top_fe(F, a, len, &r); bind_top_fe(F, b, len, &r); bind_top_fe(F, c, len, &r); eq_bound(F, q, &r);
    len /= 2;
  }
  if (rc == 0 && out_claims) { st_canon(F, out_claims, &a[0]); st_canon(F, out_claims + 32, &b[0]); st_canon(F, out_claims + 64, &c[0]); }
  eq_free(q); free(a); free(b); free(c);
  return rc;
}

/* SumcheckProof::prove_quad_prod (src/spartan/sumcheck.rs:199-249) with compute_eval_points_quad_prod (:163-186).
 * out_polys: num_rounds x 3 coefficients; out_claims: [A(r), B(r)]. */
int ref_sumcheck_prove_quad_prod(int field, const uint8_t* claim, size_t num_rounds, const uint8_t* A, const uint8_t* B, ref_transcript_fn cb,
                                 void* ctx, uint8_t* out_polys, uint8_t* out_r, uint8_t* out_claims) {
  const field_t* F = field_by_id(field); if (!F) return -1;
  size_t len = (size_t)1 << num_rounds;
  fe *a = load_vec_mont(F, A, len), *b = load_vec_mont(F, B, len);
  fe claim_per_round; ld_mont(F, &claim_per_round, claim);
  int rc = 0, T = nthreads();
  fe* part = (fe*)calloc(2 * (size_t)T, sizeof(fe));
  for (size_t j = 0; j < num_rounds && rc == 0; j++) {
    size_t h = len / 2;
    memset(part, 0, 2 * (size_t)T * sizeof(fe));
#pragma omp parallel num_threads(T)
    {
#ifdef _OPENMP
      int t = omp_get_thread_num(), nt = omp_get_num_threads();
#else
      int t = 0, nt = 1;
#endif
      fe s0, s1; memset(&s0, 0, sizeof s0); memset(&s1, 0, sizeof s1);
      for (size_t i = h * t / nt; i < h * (t + 1) / nt; i++) {
        fe x, da, db; fe_mul(F, &x, &a[i], &b[i]); fe_add(F, &s0, &s0, &x);
        fe_sub(F, &da, &a[i + h], &a[i]); fe_sub(F, &db, &b[i + h], &b[i]); fe_mul(F, &x, &da, &db); fe_add(F, &s1, &s1, &x);
      }
      part[2 * t] = s0; part[2 * t + 1] = s1;
    }
    fe e0, bc; memset(&e0, 0, sizeof e0); memset(&bc, 0, sizeof bc);
    for (int t = 0; t < T; t++) { fe_add(F, &e0, &e0, &part[2 * t]); fe_add(F, &bc, &bc, &part[2 * t + 1]); }
    fe co[3], s1;                                              /* from_evals_deg2([e0, claim - e0, bc]) */
    fe_sub(F, &s1, &claim_per_round, &e0);
    unipoly_from_evals_deg2(F, &e0, &s1, &bc, co);
    fe r;
    rc = ask_challenge(F, cb, ctx, co, 3, &r, out_polys ? out_polys + 96 * j : NULL, out_r ? out_r + 32 * j : NULL);
    if (rc) break;
    unipoly_eval(F, &claim_per_round, co, 3, &r);
    bind_top_fe(F, a, len, &r); bind_top_fe(F, b, len, &r);
    len = h;
  }
  if (rc == 0 && out_claims) { st_canon(F, out_claims, &a[0]); st_canon(F, out_claims + 32, &b[0]); }
  free(part); free(a); free(b);
  return rc;
}

/* SumcheckProof::prove_batch_eval (src/spartan/sumcheck.rs:251-353): k claims e_i = sum_x P_i(x) eq(x_i, x), polynomial i over
 * num_rounds[i] variables.  polys[i]: 2^num_rounds[i] canonical elements; eq_points[i]: num_rounds[i] elements; coeffs: k
 * elements.  out_polys: max-rounds x 3 coefficients; out_finals: [P_i(r_suffix)]. */
int ref_sumcheck_prove_batch_eval(int field, const uint8_t* claims, const size_t* num_rounds, const uint8_t* const* polys,
                                  const uint8_t* const* eq_points, const uint8_t* coeffs, size_t k, ref_transcript_fn cb, void* ctx,
                                  uint8_t* out_polys, uint8_t* out_r, uint8_t* out_finals) {
  const field_t* F = field_by_id(field); if (!F || k == 0) return -1;
  size_t nmax = 0;
  for (size_t i = 0; i < k; i++) if (num_rounds[i] > nmax) nmax = num_rounds[i];
  fe** P = (fe**)malloc(k * sizeof(fe*));
  eq_inst** Q = (eq_inst**)malloc(k * sizeof(eq_inst*));
  size_t* len = (size_t*)malloc(k * sizeof(size_t));
  fe *cl = load_vec_mont(F, claims, k), *run = load_vec_mont(F, claims, k), *co = load_vec_mont(F, coeffs, k);
  for (size_t i = 0; i < k; i++) { len[i] = (size_t)1 << num_rounds[i]; P[i] = load_vec_mont(F, polys[i], len[i]); Q[i] = eq_new(F, eq_points[i], num_rounds[i]); }
  fe e; memset(&e, 0, sizeof e);
  for (size_t i = 0; i < k; i++) {                             /* (:281-289) e = sum claim_i 2^(nmax - n_i) coeff_i */
    fe sc, t; fe_pow2(F, &sc, nmax - num_rounds[i]); fe_mul(F, &t, &cl[i], &sc); fe_mul(F, &t, &t, &co[i]); fe_add(F, &e, &e, &t);
  }
  const fe tinv = two_inv(F);
  fe (*ev)[3] = (fe(*)[3])malloc(k * sizeof(fe[3]));
  int rc = 0;
  for (size_t round = 0; round < nmax && rc == 0; round++) {
    size_t remaining = nmax - round;
    for (size_t i = 0; i < k; i++) {
      if (remaining <= num_rounds[i]) {                        /* (:301-305) */
        fe lead; eq_eval_points(F, Q[i], 1, P[i], NULL, NULL, len[i], &run[i], &ev[i][0], &lead, &ev[i][2]);
        memset(&ev[i][1], 0, sizeof(fe));
      } else {                                                 /* not yet started: constant (:306-312) */
        fe sc; fe_pow2(F, &sc, remaining - num_rounds[i] - 1); fe_mul(F, &ev[i][0], &sc, &cl[i]);
        memset(&ev[i][1], 0, sizeof(fe)); ev[i][2] = ev[i][0];
      }
    }
    fe c0, cm1, c1, qc, t; memset(&c0, 0, sizeof c0); memset(&cm1, 0, sizeof cm1);
    for (size_t i = 0; i < k; i++) { fe_mul(F, &t, &ev[i][0], &co[i]); fe_add(F, &c0, &c0, &t); fe_mul(F, &t, &ev[i][2], &co[i]); fe_add(F, &cm1, &cm1, &t); }
    fe_sub(F, &c1, &e, &c0);
    fe_add(F, &qc, &c1, &cm1); fe_dbl(F, &t, &c0); fe_sub(F, &qc, &qc, &t); fe_mul(F, &qc, &qc, &tinv);   /* (S(1) + S(-1) - 2 S(0)) / 2 */
    fe poly[3];
    unipoly_from_evals_deg2(F, &c0, &c1, &qc, poly);
    fe r;
    rc = ask_challenge(F, cb, ctx, poly, 3, &r, out_polys ? out_polys + 96 * round : NULL, out_r ? out_r + 32 * round : NULL);
    if (rc) break;
    for (size_t i = 0; i < k; i++) {
      if (remaining <= num_rounds[i]) {
        /* update_claim (:68-75) with evals = [e0, c3 = 0, em1]: a1 = (e1 - em1)/2 - c3, a2 = (e1 + em1)/2 - e0,
         * claim' = e0 + r (a1 + r (a2 + r c3)) */
        fe s0 = ev[i][0], sm1 = ev[i][2], s1, b2, cc[3];
        fe_sub(F, &s1, &run[i], &s0);
        fe_sub(F, &b2, &s1, &sm1); fe_mul(F, &cc[1], &b2, &tinv);
        fe_add(F, &b2, &s1, &sm1); fe_mul(F, &cc[2], &b2, &tinv); fe_sub(F, &cc[2], &cc[2], &s0);
        cc[0] = s0;
        unipoly_eval(F, &run[i], cc, 3, &r);
        bind_top_fe(F, P[i], len[i], &r); len[i] /= 2; eq_bound(F, Q[i], &r);
      }
    }
    unipoly_eval(F, &e, poly, 3, &r);
  }
  if (rc == 0 && out_finals) for (size_t i = 0; i < k; i++) st_canon(F, out_finals + 32 * i, &P[i][0]);
  for (size_t i = 0; i < k; i++) { free(P[i]); eq_free(Q[i]); }
  free(P); free(Q); free(len); free(cl); free(run); free(co); free(ev);
  return rc;
}

/* batch_invert_serial (src/spartan/mod.rs:120-152; batch_invert :54-118 is the same trick chunked over threads): Montgomery's trick;
 * returns 1 when an element is zero (Err(NovaError::InternalError)), canonical in / out */
int ref_batch_invert(int field, const uint8_t* v, size_t n, uint8_t* out) {
  const field_t* F = field_by_id(field); if (!F) return -1;
  fe* x = load_vec_mont(F, v, n);
  fe* products = (fe*)malloc((n ? n : 1) * sizeof(fe));
  fe acc = F->r1;
  for (size_t i = 0; i < n; i++) { products[i] = acc; fe_mul(F, &acc, &acc, &x[i]); }
  if (fe_is_zero(&acc)) { free(x); free(products); return 1; }
  fe inv; fe_inv(F, &inv, &acc);
  for (size_t i = n; i-- > 0;) {
    fe t; fe_mul(F, &t, &products[i], &inv); fe_mul(F, &inv, &inv, &x[i]);
    st_canon(F, out + 32 * i, &t);
  }
  free(x); free(products);
  return 0;
}

/* ------------------------------------------------------------------ inner-product argument ---------------------
 * InnerProductArgument::prove (src/provider/ipa_pc.rs:174-281) behind EvaluationEngine::prove (:69-82; S2 of CompressedSNARK::prove
 * on the secondary curve, src/nova/mod.rs:862-881), restated step by step INCLUDING the key fold the HIP path avoids:
 *   per round (prove_inner, :195-251):  (ck_L, ck_R) = ck.split_at(n/2)                      pedersen.rs:457-468
 *     c_L = <a[..n/2], b[n/2..]>, c_R = <a[n/2..], b[..n/2]>                                  inner_product, :84-90
 *     L = commit(ck_R.combine(ck_c), a[..n/2] || c_L, 0), R = commit(ck_L.combine(ck_c), a[n/2..] || c_R, 0)   :213-232
 *     r = transcript(L, R); a' = a_L r + r^-1 a_R; b' = b_L r^-1 + r b_R                       :234-249
 *     ck' = ck.fold(r^-1, r): ck'[i] = msm([r^-1, r], [ck_L[i], ck_R[i]]).affine()             pedersen.rs:484-497
 *   result: L_vec, R_vec, a_hat = a[0]                                                         :268-272
 * `ck_c` is the ALREADY SCALED one-point key (ck_c.scale(&r), :190-191: the caller's transcript produced that r); the callback
 * absorbs L and R and squeezes the round's r (:231-234), returning it canonical.  n must be a power of two (the evaluation engine
 * passes 2^ell evaluations).  Returns -1 on bad arguments, -11 if a challenge is zero (`r.invert().unwrap()` panics). */
typedef int (*ref_ipa_transcript_fn)(void* ctx, const uint8_t* L_xy64, int L_is_inf, const uint8_t* R_xy64, int R_is_inf, uint8_t* r32);
int ref_ipa_prove(int curve, const uint8_t* ck_xy64, const uint8_t* ck_c_xy64, const uint8_t* a_le32, const uint8_t* b_le32, size_t n,
                  ref_ipa_transcript_fn cb, void* ctx, uint8_t* out_L, uint8_t* out_R, uint8_t* out_inf, uint8_t* out_a_hat) {
  if (curve < 0 || curve > 3 || n == 0 || (n & (n - 1))) return -1;
  const curve_t* C = &CURVES[curve];
  const field_t* S = C->scalar;
  for (size_t i = 0; i < n; i++) {
    fe s; memcpy(&s, a_le32 + 32 * i, 32); if (fe_geq(&s, &S->p)) return -4;
    memcpy(&s, b_le32 + 32 * i, 32); if (fe_geq(&s, &S->p)) return -4;
  }
  aff* ck = (aff*)malloc(sizeof(aff) * (n + 1));       /* Montgomery coordinates, as a host Vec<Affine> */
  aff* key = (aff*)malloc(sizeof(aff) * (n / 2 + 2));  /* ck_R.combine(ck_c) / ck_L.combine(ck_c) */
  fe* a = load_vec_mont(S, a_le32, n);
  fe* b = load_vec_mont(S, b_le32, n);
  fe* v = (fe*)malloc(sizeof(fe) * (n / 2 + 2));       /* canonical scalars of one commitment */
  load_bases(C, ck_xy64, n, ck);
  aff ckc; load_bases(C, ck_c_xy64, 1, &ckc);
  int rc = 0;
  size_t round = 0;
  for (size_t len = n; len > 1 && rc == 0; len /= 2, round++) {
    const size_t h = len / 2;
    fe cL, cR; memset(&cL, 0, sizeof cL); memset(&cR, 0, sizeof cR);
    for (size_t i = 0; i < h; i++) {
      fe t; fe_mul(S, &t, &a[i], &b[h + i]); fe_add(S, &cL, &cL, &t);
      fe_mul(S, &t, &a[h + i], &b[i]); fe_add(S, &cR, &cR, &t);
    }
    xyzz acc;
    uint8_t Lb[64], Rb[64], Li = 0, Ri = 0, ch[32];
    /* L: bases ck_R || ck_c, scalars a_L || c_L; the blinding term is h * 0 (commit(.., &Scalar::ZERO)) */
    memcpy(key, ck + h, sizeof(aff) * h); key[h] = ckc;
    for (size_t i = 0; i < h; i++) fe_from_mont(S, &v[i], &a[i]);
    fe_from_mont(S, &v[h], &cL);
    msm_full(C, v, key, h + 1, &acc); store_point(C, &acc, Lb, &Li);
    memcpy(key, ck, sizeof(aff) * h); key[h] = ckc;
    for (size_t i = 0; i < h; i++) fe_from_mont(S, &v[i], &a[h + i]);
    fe_from_mont(S, &v[h], &cR);
    msm_full(C, v, key, h + 1, &acc); store_point(C, &acc, Rb, &Ri);
    memcpy(out_L + 64 * round, Lb, 64); memcpy(out_R + 64 * round, Rb, 64);
    if (out_inf) { out_inf[2 * round] = Li; out_inf[2 * round + 1] = Ri; }
    if (cb(ctx, Lb, Li, Rb, Ri, ch) != 0) { rc = -1; break; }
    fe rr; memcpy(&rr, ch, 32);
    if (fe_geq(&rr, &S->p)) { rc = -4; break; }
    if (fe_is_zero(&rr)) { rc = -11; break; }
    fe r, rinv; fe_to_mont(S, &r, &rr); fe_inv(S, &rinv, &r);
#pragma omp parallel for num_threads(nthreads())
    for (long i = 0; i < (long)h; i++) {
      fe t, u;
      fe_mul(S, &t, &a[i], &r); fe_mul(S, &u, &rinv, &a[h + i]); fe_add(S, &a[i], &t, &u);
      fe_mul(S, &t, &b[i], &rinv); fe_mul(S, &u, &r, &b[h + i]); fe_add(S, &b[i], &t, &u);
    }
    /* ck.fold(&r_inverse, &r): one two-point MSM per pair, result normalised to affine (pedersen.rs:488-494) */
    fe w[2]; fe_from_mont(S, &w[0], &rinv); fe_from_mont(S, &w[1], &r);
#pragma omp parallel for num_threads(nthreads()) schedule(dynamic, 16)
    for (long i = 0; i < (long)h; i++) {
      aff two[2] = {ck[i], ck[h + i]};
      xyzz t; msm_full(C, w, two, 2, &t);
      aff o; xyzz_to_affine(C->base, &o, &t);
      ck[i] = o;   /* ck[i] is read by iteration i only; ck[h + i] is never written in this loop */
    }
  }
  if (rc == 0) st_canon(S, out_a_hat, &a[0]);
  free(ck); free(key); free(a); free(b); free(v);
  return rc;
}
