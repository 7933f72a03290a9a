// tests/cpp/host_fp4_test.cpp -- g++-only check of nova_amd/csrc/host_fp4.hpp (the provers' host-side field arithmetic): prints
// operands and results as hex lines that tests/test_host_fp4.py verifies with Python integers.
#include <stdio.h>
#include <stdlib.h>

#include "../../nova_amd/csrc/host_fp4.hpp"

using namespace nmx;

static uint64_t s = 0x5EEDC0DE12345678ull;
static uint64_t rnd() {
  s += 0x9e3779b97f4a7c15ull;
  uint64_t x = s;
  x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
  x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
  return x ^ (x >> 31);
}
static void hex(const char* name, const void* w32) {
  const uint8_t* b = (const uint8_t*)w32;
  printf("%s=", name);
  for (int i = 31; i >= 0; i--) printf("%02x", b[i]);
  printf(" ");
}
template <int FID> void run(int cases) {
  using H = HostFp4<FID>;
  for (int t = 0; t < cases; t++) {
    uint64_t a[4], b[4];
    for (int i = 0; i < 4; i++) a[i] = rnd(), b[i] = rnd();
    a[3] &= 0x0fffffffffffffffull, b[3] &= 0x0fffffffffffffffull;  // < 2^252 < p
    if (t == 0) a[0] = a[1] = a[2] = a[3] = 0;
    if (t == 1) a[0] = 1, a[1] = a[2] = a[3] = 0;
    const H x = H::from_canonical(a), y = H::from_canonical(b);
    uint8_t o[32];
    printf("fid=%d ", FID);
    hex("a", a), hex("b", b);
    (x * y).to_canonical(o), hex("mul", o);
    (x + y).to_canonical(o), hex("add", o);
    (x - y).to_canonical(o), hex("sub", o);
    x.inv().to_canonical(o), hex("inv", o);
    x.to_mont256(o), hex("mont", o);
    H::from_mont256(o).to_canonical(o), hex("back", o);
    H::from_plain_times(a, H::pow2(t % 300)).to_canonical(o), hex("pt", o);
    printf("e=%d ", t % 300);
    uint32_t w[8];
    x.to_device().canon().to_words(w), hex("dev", w);
    // the device form's own conversion of the same canonical integer must agree
    uint32_t aw[8];
    memcpy(aw, a, 32);
    Fp<FID>::from_words(aw).to_internal().canon().to_words(w), hex("dev_ref", w);
    printf("\n");
  }
}
// the two inversions (31 steps at a time on approximations; bit at a time) on values chosen to stress the approximation: small,
// p - small, around powers of two, long runs of zero bits, random of every length.  "fast" = 1: the fast path converged (its
// answer is then what inv() returned after checking it); every line also carries the slow path's answer.
template <int FID> void run_inv(int cases) {
  using H = HostFp4<FID>;
  auto emit = [](const H& x) {
    uint8_t o[32];
    uint64_t z4[4];
    printf("invfid=%d fast=%d ", FID, (!x.is_zero() && H::inv_plain(x.v, z4)) ? 1 : 0);
    x.to_canonical(o), hex("a", o);
    x.inv().to_canonical(o), hex("inv", o);
    x.inv_slow().to_canonical(o), hex("inv_slow", o);
    printf("\n");
  };
  emit(H::zero());
  for (uint64_t k = 1; k <= 40; k++) emit(H::from_u64(k)), emit(H::zero() - H::from_u64(k));
  for (uint32_t e = 0; e < 520; e += 3) emit(H::pow2(e)), emit(H::pow2(e) - H::one()), emit(H::pow2(e) + H::from_u64(3));
  for (int t = 0; t < cases; t++) {
    uint64_t a[4] = {rnd(), rnd(), rnd(), rnd() & 0x0fffffffffffffffull};
    const int keep = t % 252 + 1;  // a random value of `keep` bits
    for (int i = 0; i < 4; i++) {
      const int lo = 64 * i;
      if (keep <= lo) a[i] = 0;
      else if (keep < lo + 64) a[i] &= (((uint64_t)1 << (keep - lo)) - 1);
    }
    if (t % 3 == 0) a[1] = 0;  // a long run of zero bits inside
    if (!(a[0] | a[1] | a[2] | a[3])) a[0] = 5;
    emit(H::from_canonical(a));
  }
}
int main(int argc, char** argv) {
  const int cases = argc > 1 ? atoi(argv[1]) : 50;
  run<0>(cases), run<1>(cases), run<2>(cases), run<3>(cases);
  run_inv<0>(20 * cases), run_inv<1>(20 * cases), run_inv<2>(20 * cases), run_inv<3>(20 * cases);
  return 0;
}
