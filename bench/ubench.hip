// bench/ubench.hip -- instruction-rate microbenchmarks that decide the field-arithmetic design on gfx950.
// The hardware guide gives no integer-multiply rate; this measures it, next to the FP64 FMA
// rate (the alternative "DPF" big-integer multiplier) and the modmul throughput of candidate formulations.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o bench/ubench bench/ubench.hip && bench/ubench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
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

constexpr int ITERS = 2048;

__global__ __launch_bounds__(256) void k_mad64(uint64_t* out, uint32_t a, uint32_t b) {
  uint64_t x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
  uint32_t m = a + threadIdx.x, n = b;
  for (int i = 0; i < ITERS; i++) {
    x0 = (uint64_t)m * n + x0;
    x1 = (uint64_t)m * n + x1;
    x2 = (uint64_t)m * n + x2;
    x3 = (uint64_t)m * n + x3;
    m = (uint32_t)x0;
    n = (uint32_t)x1;
  }
  out[blockIdx.x * 256 + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3;
}
__global__ __launch_bounds__(256) void k_mullo(uint32_t* out, uint32_t a) {
  uint32_t x0 = threadIdx.x | 1, x1 = x0 + 2, x2 = x0 + 4, x3 = x0 + 6;
  for (int i = 0; i < ITERS; i++) {
    x0 = x0 * a + 1;
    x1 = x1 * a + 1;
    x2 = x2 * a + 1;
    x3 = x3 * a + 1;
  }
  out[blockIdx.x * 256 + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3;
}
__global__ __launch_bounds__(256) void k_mulhi(uint32_t* out, uint32_t a) {
  uint32_t x0 = threadIdx.x | 0x80000001u, x1 = x0 + 2, x2 = x0 + 4, x3 = x0 + 6;
  for (int i = 0; i < ITERS; i++) {
    x0 = __umulhi(x0, a) | 0x80000000u;
    x1 = __umulhi(x1, a) | 0x80000000u;
    x2 = __umulhi(x2, a) | 0x80000000u;
    x3 = __umulhi(x3, a) | 0x80000000u;
  }
  out[blockIdx.x * 256 + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3;
}
__global__ __launch_bounds__(256) void k_add32(uint32_t* out, uint32_t a) {
  uint32_t x0 = threadIdx.x, x1 = x0 + 2, x2 = x0 + 4, x3 = x0 + 6;
  for (int i = 0; i < ITERS; i++) {
    x0 = (x0 + a) ^ x1;
    x1 = (x1 + a) ^ x2;
    x2 = (x2 + a) ^ x3;
    x3 = (x3 + a) ^ x0;
  }
  out[blockIdx.x * 256 + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3;
}
__global__ __launch_bounds__(256) void k_dfma(double* out, double a, double b) {
  double x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
  for (int i = 0; i < ITERS; i++) {
    x0 = __builtin_fma(x0, a, b);
    x1 = __builtin_fma(x1, a, b);
    x2 = __builtin_fma(x2, a, b);
    x3 = __builtin_fma(x3, a, b);
  }
  out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3;
}

// --- modmul variants -------------------------------------------------------------------------------
// V0: fp.hpp as shipped (C++ 96-bit accumulator)
__global__ __launch_bounds__(256) void k_modmul_v0(Fp<0>* io, int iters) {
  int t = blockIdx.x * 256 + threadIdx.x;
  Fp<0> x = io[t], y = io[t + 1];
  for (int i = 0; i < iters; i++) {
    x = x * y;
    y = y * x;
  }
  io[t] = x + y;
}

// V1: asm mac (v_mad_u64_u32 carry-out -> v_addc), 2 wait states for the VCC hazard
__device__ __forceinline__ void mac_asm(uint64_t& lo, uint32_t& hi, uint32_t x, uint32_t y) {
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\ts_nop 1\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc"
      : "+v"(lo), "+v"(hi)
      : "v"(x), "v"(y)
      : "vcc");
}
__device__ __forceinline__ Fp<0> mul_v1(const Fp<0>& a, const Fp<0>& b) {
  using PP = FpParams<0>;
  uint64_t lo = 0;
  uint32_t hi = 0;
  uint32_t m[8];
  Fp<0> r;
#pragma unroll
  for (int k = 0; k < 8; k++) {
#pragma unroll
    for (int i = 0; i <= k; i++) mac_asm(lo, hi, a.l[i], b.l[k - i]);
#pragma unroll
    for (int i = 0; i < k; i++) mac_asm(lo, hi, m[i], PP::P[k - i]);
    m[k] = (uint32_t)lo * PP::NINV;
    mac_asm(lo, hi, m[k], PP::P[0]);
    lo = (lo >> 32) | ((uint64_t)hi << 32);
    hi = 0;
  }
#pragma unroll
  for (int k = 8; k < 16; k++) {
#pragma unroll
    for (int i = k - 7; i < 8; i++) mac_asm(lo, hi, a.l[i], b.l[k - i]);
#pragma unroll
    for (int i = k - 7; i < 8; i++) mac_asm(lo, hi, m[i], PP::P[k - i]);
    r.l[k - 8] = (uint32_t)lo;
    lo = (lo >> 32) | ((uint64_t)hi << 32);
    hi = 0;
  }
  r.cond_sub_p();
  return r;
}
__global__ __launch_bounds__(256) void k_modmul_v1(Fp<0>* io, int iters) {
  int t = blockIdx.x * 256 + threadIdx.x;
  Fp<0> x = io[t], y = io[t + 1];
  for (int i = 0; i < iters; i++) {
    x = mul_v1(x, y);
    y = mul_v1(y, x);
  }
  io[t] = x + y;
}

// V2: 9 x 29-bit limbs, plain 64-bit column accumulators (no carry logic at all); throughput probe only
struct F29 {
  uint32_t l[9];
};
__device__ __forceinline__ F29 mul_v2(const F29& a, const F29& b, const F29& p, uint32_t ninv) {
  const uint32_t mask = (1u << 29) - 1;
  uint64_t acc = 0;
  uint32_t m[9];
  F29 r;
#pragma unroll
  for (int k = 0; k < 9; k++) {
#pragma unroll
    for (int i = 0; i <= k; i++) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
    for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * p.l[k - i];
    m[k] = ((uint32_t)acc * ninv) & mask;
    acc += (uint64_t)m[k] * p.l[0];
    acc >>= 29;
  }
#pragma unroll
  for (int k = 9; k < 17; k++) {
#pragma unroll
    for (int i = k - 8; i < 9; i++) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
    for (int i = k - 8; i < 9; i++) acc += (uint64_t)m[i] * p.l[k - i];
    r.l[k - 9] = (uint32_t)acc & mask;
    acc >>= 29;
  }
  r.l[8] = (uint32_t)acc;
  return r;
}
__global__ __launch_bounds__(256) void k_modmul_v2(F29* io, int iters, F29 p, uint32_t ninv) {
  int t = blockIdx.x * 256 + threadIdx.x;
  F29 x = io[t], y = io[t + 1];
  for (int i = 0; i < iters; i++) {
    x = mul_v2(x, y, p, ninv);
    y = mul_v2(y, x, p, ninv);
  }
#pragma unroll
  for (int j = 0; j < 9; j++) x.l[j] ^= y.l[j];
  io[t] = x;
}

// V3: operand-scanning CIOS on 32-bit limbs, every step  a*b + t + c  (cannot overflow 64 bits: no carry flags)
__device__ __forceinline__ Fp<0> mul_v3(const Fp<0>& a, const Fp<0>& b) {
  using PP = FpParams<0>;
  uint32_t t[9];
#pragma unroll
  for (int j = 0; j < 9; j++) t[j] = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    uint64_t c = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      c = (uint64_t)a.l[j] * b.l[i] + t[j] + c;
      t[j] = (uint32_t)c;
      c >>= 32;
    }
    uint32_t t8 = t[8] + (uint32_t)c;  // < 2^32 because P < 2^255
    uint32_t m = t[0] * PP::NINV;
    c = (uint64_t)m * PP::P[0] + t[0];
    c >>= 32;
#pragma unroll
    for (int j = 1; j < 8; j++) {
      c = (uint64_t)m * PP::P[j] + t[j] + c;
      t[j - 1] = (uint32_t)c;
      c >>= 32;
    }
    c += t8;
    t[7] = (uint32_t)c;
    t[8] = (uint32_t)(c >> 32);
  }
  Fp<0> r;
#pragma unroll
  for (int j = 0; j < 8; j++) r.l[j] = t[j];
  r.cond_sub_p();
  return r;
}
__global__ __launch_bounds__(256) void k_modmul_v3(Fp<0>* io, int iters) {
  int t = blockIdx.x * 256 + threadIdx.x;
  Fp<0> x = io[t], y = io[t + 1];
  for (int i = 0; i < iters; i++) {
    x = mul_v3(x, y);
    y = mul_v3(y, x);
  }
  io[t] = x + y;
}
// correctness cross-check of V1/V3 against V0 on random inputs
__global__ void k_check(const Fp<0>* in, uint32_t* bad, int n) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  Fp<0> a = in[t], b = in[t + 1];
  Fp<0> r0 = a * b, r1 = mul_v1(a, b), r3 = mul_v3(a, b);
  if (r0 != r1) atomicAdd(&bad[0], 1);
  if (r0 != r3) atomicAdd(&bad[1], 1);
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
  printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %d}\n", prop.name, prop.multiProcessorCount,
         prop.clockRate / 1000);
  const int blocks = prop.multiProcessorCount * 8, threads = blocks * 256;
  void* buf;
  CHK(hipMalloc(&buf, (size_t)(threads + 1) * 64));
  CHK(hipMemset(buf, 0x11, (size_t)(threads + 1) * 64));
  const double lanes_ops = (double)threads * ITERS * 4;

  float ms = time_ms([&] { k_mad64<<<blocks, 256>>>((uint64_t*)buf, 12345u, 6789u); });
  printf("{\"ubench\": \"v_mad_u64_u32\", \"Gops\": %.1f}\n", lanes_ops / ms * 1e-6);
  ms = time_ms([&] { k_mullo<<<blocks, 256>>>((uint32_t*)buf, 2654435761u); });
  printf("{\"ubench\": \"v_mul_lo_u32(+add)\", \"Gops\": %.1f}\n", lanes_ops / ms * 1e-6);
  ms = time_ms([&] { k_mulhi<<<blocks, 256>>>((uint32_t*)buf, 2654435761u); });
  printf("{\"ubench\": \"v_mul_hi_u32(+or)\", \"Gops\": %.1f}\n", lanes_ops / ms * 1e-6);
  ms = time_ms([&] { k_add32<<<blocks, 256>>>((uint32_t*)buf, 77u); });
  printf("{\"ubench\": \"v_add_u32+v_xor\", \"Gops\": %.1f}\n", 2 * lanes_ops / ms * 1e-6);
  ms = time_ms([&] { k_dfma<<<blocks, 256>>>((double*)buf, 1.0000001, 1e-9); });
  printf("{\"ubench\": \"v_fma_f64\", \"Gops\": %.1f}\n", lanes_ops / ms * 1e-6);

  // random-ish field elements < P: clear the top 3 bits
  {
    std::vector<uint32_t> h((size_t)(threads + 1) * 8);
    uint64_t s = 0x5EEDC0DE12345678ull;
    for (auto& v : h) {
      s = s * 6364136223846793005ull + 1442695040888963407ull;
      v = (uint32_t)(s >> 32);
    }
    for (size_t i = 0; i < (size_t)threads + 1; i++) h[i * 8 + 7] &= 0x1fffffffu;
    CHK(hipMemcpy(buf, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  }
  uint32_t* bad;
  CHK(hipMalloc(&bad, 8));
  CHK(hipMemset(bad, 0, 8));
  k_check<<<(threads + 255) / 256, 256>>>((const Fp<0>*)buf, bad, threads);
  uint32_t hbad[2];
  CHK(hipMemcpy(hbad, bad, 8, hipMemcpyDeviceToHost));
  printf("{\"check\": \"modmul variants vs v0\", \"v1_mismatch\": %u, \"v3_mismatch\": %u}\n", hbad[0], hbad[1]);

  const int it = 256;
  const double mm = (double)threads * it * 2;
  ms = time_ms([&] { k_modmul_v0<<<blocks, 256>>>((Fp<0>*)buf, it); });
  printf("{\"ubench\": \"modmul_v0_cpp96\", \"Gmodmul_s\": %.2f}\n", mm / ms * 1e-6);
  ms = time_ms([&] { k_modmul_v1<<<blocks, 256>>>((Fp<0>*)buf, it); });
  printf("{\"ubench\": \"modmul_v1_asm_mac\", \"Gmodmul_s\": %.2f}\n", mm / ms * 1e-6);
  ms = time_ms([&] { k_modmul_v3<<<blocks, 256>>>((Fp<0>*)buf, it); });
  printf("{\"ubench\": \"modmul_v3_cios\", \"Gmodmul_s\": %.2f}\n", mm / ms * 1e-6);
  F29 p29;
  for (int j = 0; j < 9; j++) p29.l[j] = 0x0fffffffu - j;
  p29.l[0] |= 1;
  ms = time_ms([&] { k_modmul_v2<<<blocks, 256>>>((F29*)buf, it, p29, 0x12345677u); });
  printf("{\"ubench\": \"modmul_v2_29bit\", \"Gmodmul_s\": %.2f}\n", mm / ms * 1e-6);
  return 0;
}
