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
int main(int argc, char** argv) {
  const int cases = argc > 1 ? atoi(argv[1]) : 50;
  run<0>(cases), run<1>(cases), run<2>(cases), run<3>(cases);
  return 0;
}
