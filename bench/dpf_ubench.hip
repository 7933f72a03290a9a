// bench/dpf_ubench.hip -- settles "would a double-precision-FMA multiplier be cheaper than the 9 x 29-bit v_mad_u64_u32 chain?"
// by measurement (VERDICT r4 next #7a; DESIGN.md section 5 argued it from rates alone).
//
// The DPF formulation (Emmart, Zheng, Weems: "Faster modular exponentiation using double precision floating point arithmetic
// on the GPU", ARITH 2018): 52-bit limbs held in doubles; a limb product a b < 2^104 is split by two FMAs in round-toward-zero
//     hi = fma(a, b, 2^104)          = 2^104 + 2^52 floor(a b / 2^52)          (the mantissa IS the high half)
//     lo = fma(a, b, (2^104 + 2^52) - hi) = 2^52 + (a b mod 2^52)              (the mantissa IS the low half)
// and the halves are accumulated column by column as 64-bit INTEGER adds of the raw bit patterns (the constants' patterns are
// subtracted once per column).  A 256-bit operand is 5 limbs: 25 limb products for the schoolbook half of a Montgomery product
// against 81 for 9 x 29 bits -- but each costs 2 FMAs + 1 FP64 add + 2 64-bit integer adds instead of ONE v_mad_u64_u32 (which
// multiplies AND accumulates 64 bits), and v_fma_f64 issues at the same rate as v_mad_u64_u32 (profiles/r01_ubench_*.jsonl: 31.3
// against 29.0 T lane-ops/s).  This file times exactly that schoolbook half -- 5 x 5 DPF product with column accumulation,
// constant removal, carry propagation back to 52-bit limbs and the re-entry into doubles, verified bit for bit against
// integer arithmetic on the host -- against the shipped 9 x 29 modular product (fp.hpp), whose schoolbook half is half of it.
// A full DPF Montgomery product is this twice (a b, then q p) plus the q limbs (5 more low products): DPF wins only if
// 2 x t(5x5 DPF) < t(9x29 modmul).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o bench/dpf_ubench bench/dpf_ubench.hip && bench/dpf_ubench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../nova_amd/csrc/fp.hpp"

using namespace nmx;
#define CHK(x)                                                                      \
  do {                                                                              \
    hipError_t e = (x);                                                             \
    if (e != hipSuccess) {                                                          \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__);  \
      exit(1);                                                                      \
    }                                                                               \
  } while (0)

static constexpr uint64_t M52 = (1ull << 52) - 1;
// FP64 rounding mode (MODE register bits [3:2]) <- toward zero / back to nearest.  The operands pass THROUGH the asm statement:
// the compiler knows nothing about the mode register and would otherwise be free to schedule the FMAs ahead of the switch (the
// first build of this file did exactly that: 198 of 200 products wrong, profiles/r05_ubench_dpf.jsonl).
__device__ __forceinline__ void round_toward_zero(double (&x)[5], double (&y)[5]) {
  asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 3\n\ts_nop 1"
               : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(y[0]), "+v"(y[1]), "+v"(y[2]), "+v"(y[3]), "+v"(y[4]));
}
__device__ __forceinline__ void round_to_nearest(uint64_t (&p)[10]) {
  asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 0\n\ts_nop 1"
               : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]), "+v"(p[8]), "+v"(p[9]));
}
__device__ __forceinline__ uint64_t dbits(double d) { return (uint64_t)__double_as_longlong(d); }
__device__ __forceinline__ double limb_to_double(uint64_t l) {  // 52-bit integer -> double: OR into the mantissa of 2^52, subtract 2^52
  return __longlong_as_double((long long)(l | 0x4330000000000000ull)) - 4503599627370496.0;
}
// out[0..10) = a * b as ten 52-bit limbs; a, b: five limbs in doubles.  FP64 rounding mode must be toward zero.
__device__ __forceinline__ void dpf_mul5(const double (&a)[5], const double (&b)[5], uint64_t (&out)[10]) {
  const double C1 = 20282409603651670423947251286016.0;              // 2^104
  const double C2 = 20282409603651670423947251286016.0 + 4503599627370496.0;  // 2^104 + 2^52
  uint64_t hi[10], lo[10];
#pragma unroll
  for (int k = 0; k < 10; k++) hi[k] = lo[k] = 0;
#pragma unroll
  for (int i = 0; i < 5; i++)
#pragma unroll
    for (int j = 0; j < 5; j++) {
      const double h = __builtin_fma(a[i], b[j], C1);
      const double t = C2 - h;
      const double l = __builtin_fma(a[i], b[j], t);
      hi[i + j + 1] += dbits(h);
      lo[i + j] += dbits(l);
    }
  // remove the constants' bit patterns: column k received cnt(k) low halves and cnt(k - 1) high halves
  const uint64_t B1 = 0x4670000000000000ull, B2 = 0x4330000000000000ull;  // bits(2^104), bits(2^52)
  uint64_t carry = 0;
#pragma unroll
  for (int k = 0; k < 10; k++) {
    const int cl = k < 5 ? k + 1 : (k < 9 ? 9 - k : 0), ch = k >= 1 ? (k - 1 < 5 ? k : 10 - k) : 0;
    uint64_t v = lo[k] - (uint64_t)cl * B2 + hi[k] - (uint64_t)ch * B1 + carry;
    out[k] = v & M52;
    carry = v >> 52;
  }
}
__global__ __launch_bounds__(256) void k_dpf_mul(uint64_t* io, int iters) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  double x[5], y[5];
#pragma unroll
  for (int i = 0; i < 5; i++) x[i] = limb_to_double(io[10 * (size_t)t + i]), y[i] = limb_to_double(io[10 * (size_t)t + 5 + i]);
  round_toward_zero(x, y);
  uint64_t p[10];
  for (int it = 0; it < iters; it++) {
    // all ten limbs feed the next operand (low half + high half, limb by limb): nothing of the 25 products is dead code, and
    // the re-entry into doubles is what a Montgomery step pays too
    dpf_mul5(x, y, p);
#pragma unroll
    for (int i = 0; i < 5; i++) x[i] = limb_to_double((p[i] + p[i + 5]) & M52);
    dpf_mul5(y, x, p);
#pragma unroll
    for (int i = 0; i < 5; i++) y[i] = limb_to_double((p[i] + p[i + 5]) & M52);
  }
#pragma unroll
  for (int i = 0; i < 5; i++) p[i] = dbits(x[i] + 4503599627370496.0) & M52, p[5 + i] = dbits(y[i] + 4503599627370496.0) & M52;
  round_to_nearest(p);
#pragma unroll
  for (int i = 0; i < 10; i++) io[10 * (size_t)t + i] = p[i];
}
// one product, all ten limbs out (verification)
__global__ void k_dpf_once(const uint64_t* in, uint64_t* out) {
  double x[5], y[5];
  for (int i = 0; i < 5; i++) x[i] = limb_to_double(in[i]), y[i] = limb_to_double(in[5 + i]);
  round_toward_zero(x, y);
  uint64_t p[10];
  dpf_mul5(x, y, p);
  round_to_nearest(p);
  for (int i = 0; i < 10; i++) out[i] = p[i];
}
template <int FID> __global__ __launch_bounds__(256) void k_modmul(uint32_t* io, int iters) {
  int t = blockIdx.x * 256 + threadIdx.x;
  Fp<FID> x = Fp<FID>::from_words(io + 8 * t), y = Fp<FID>::from_words(io + 8 * t + 8);
  for (int i = 0; i < iters; i++) {
    x = x * y;
    y = y * x;
  }
  (x + y).norm().canon().to_words(io + 8 * t);
}
template <class L> float time_ms(L&& launch, int reps = 5) {
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0));
  CHK(hipEventCreate(&e1));
  launch();
  CHK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < reps; r++) {
    CHK(hipEventRecord(e0));
    launch();
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  return best;
}
int main() {
  hipDeviceProp_t prop;
  CHK(hipGetDeviceProperties(&prop, 0));
  printf("{\"device\": \"%s\", \"cus\": %d}\n", prop.name, prop.multiProcessorCount);
  // --- verification: random 52-bit limbs, the ten-limb product against unsigned __int128 schoolbook on the host
  uint64_t s = 0x5EEDC0DE12345678ull;
  auto rnd = [&] {
    s += 0x9e3779b97f4a7c15ull;
    uint64_t x = s;
    x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
    x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
    return x ^ (x >> 31);
  };
  uint64_t *din, *dout;
  CHK(hipMalloc(&din, 80));
  CHK(hipMalloc(&dout, 80));
  int bad = 0;
  for (int c = 0; c < 200; c++) {
    uint64_t in[10], got[10], exp[10];
    for (int i = 0; i < 10; i++) in[i] = c == 0 ? M52 : (c == 1 ? 0 : rnd() & M52);  // all-ones limbs: every column at its maximum
    CHK(hipMemcpy(din, in, 80, hipMemcpyHostToDevice));
    k_dpf_once<<<1, 1>>>(din, dout);
    CHK(hipMemcpy(got, dout, 80, hipMemcpyDeviceToHost));
    unsigned __int128 col[11];
    for (auto& v : col) v = 0;
    for (int i = 0; i < 5; i++)
      for (int j = 0; j < 5; j++) {
        const unsigned __int128 pr = (unsigned __int128)in[i] * in[5 + j];
        col[i + j] += (uint64_t)(pr & M52);
        col[i + j + 1] += (uint64_t)(pr >> 52);
      }
    unsigned __int128 cy = 0;
    for (int k = 0; k < 10; k++) {
      const unsigned __int128 v = col[k] + cy;
      exp[k] = (uint64_t)(v & M52);
      cy = v >> 52;
    }
    if (memcmp(got, exp, 80) != 0) bad++;
  }
  printf("{\"check\": \"dpf 5x5 product, 200 cases incl. all-ones limbs, vs unsigned __int128 schoolbook\", \"mismatches\": %d}\n", bad);
  // --- timing
  const int blocks = prop.multiProcessorCount * 8, threads = blocks * 256, it = 256;
  void* buf;
  CHK(hipMalloc(&buf, (size_t)(threads + 1) * 80));
  {
    std::vector<uint64_t> h((size_t)(threads + 1) * 10);
    for (auto& v : h) v = rnd() & M52;
    CHK(hipMemcpy(buf, h.data(), h.size() * 8, hipMemcpyHostToDevice));
  }
  const double prods = (double)threads * it * 2;
  float ms = time_ms([&] { k_dpf_mul<<<blocks, 256>>>((uint64_t*)buf, it); });
  const double dpf_half = prods / ms * 1e-6;
  printf("{\"ubench\": \"dpf_5x52 schoolbook half (25 limb products, columns, carries, re-entry)\", \"G_per_s\": %.2f}\n", dpf_half);
  {
    std::vector<uint32_t> h((size_t)(threads + 1) * 8);
    for (auto& v : h) v = (uint32_t)rnd();
    for (size_t i = 0; i < (size_t)threads + 1; i++) h[i * 8 + 7] &= 0x1fffffffu;
    CHK(hipMemcpy(buf, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  }
  ms = time_ms([&] { k_modmul<1><<<blocks, 256>>>((uint32_t*)buf, it); });
  const double mad_full = prods / ms * 1e-6;
  printf("{\"ubench\": \"modmul_9x29 bn254_fr (the shipped product: schoolbook AND Montgomery halves)\", \"G_per_s\": %.2f}\n", mad_full);
  printf("{\"verdict\": \"a DPF Montgomery product is at least two such halves: <= %.2f G/s against %.2f G/s shipped = %.2fx\", "
         "\"keep_bar\": \">= 1.3x\"}\n", dpf_half / 2, mad_full, dpf_half / 2 / mad_full);
  return bad != 0;
}
