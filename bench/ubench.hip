// bench/ubench.hip -- instruction-rate microbenchmarks that decide the field-arithmetic design on gfx950.
// The hardware guide gives no integer-multiply rate; this measures it, next to the FP64 FMA
// rate (the alternative "DPF" big-integer multiplier) and the modmul throughput of candidate formulations.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o bench/ubench bench/ubench.hip && bench/ubench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../nova_amd/csrc/curve.hpp"

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

__global__ __launch_bounds__(256) void k_shr64_only(uint64_t* out, uint32_t a) {
  uint64_t x0 = threadIdx.x * 0x9E3779B97F4A7C15ull + a, x1 = x0 * 3, x2 = x0 * 5, x3 = x0 * 7;
  for (int i = 0; i < ITERS; i++) {
    asm volatile("v_lshrrev_b64 %0, 1, %0\n\tv_lshrrev_b64 %1, 1, %1\n\tv_lshrrev_b64 %2, 1, %2\n\tv_lshrrev_b64 %3, 1, %3"
                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
  }
  out[blockIdx.x * 256 + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3;
}
__global__ __launch_bounds__(256) void k_add64(uint64_t* out, uint32_t a) {
  uint64_t x0 = threadIdx.x + a, x1 = x0 * 3, x2 = x0 * 5, x3 = x0 * 7;
  for (int i = 0; i < ITERS; i++) {
    asm volatile("v_lshl_add_u64 %0, %1, 0, %0\n\tv_lshl_add_u64 %1, %2, 0, %1\n\tv_lshl_add_u64 %2, %3, 0, %2\n\tv_lshl_add_u64 %3, %0, 0, %3"
                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
  }
  out[blockIdx.x * 256 + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3;
}
__global__ __launch_bounds__(256) void k_alignbit(uint32_t* out, uint32_t a) {
  uint32_t x0 = threadIdx.x + a, x1 = x0 * 3, x2 = x0 * 5, x3 = x0 * 7;
  for (int i = 0; i < ITERS; i++) {
    asm volatile("v_alignbit_b32 %0, %1, %0, 29\n\tv_alignbit_b32 %1, %2, %1, 29\n\tv_alignbit_b32 %2, %3, %2, 29\n\tv_alignbit_b32 %3, %0, %3, 29"
                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
  }
  out[blockIdx.x * 256 + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3;
}
__global__ __launch_bounds__(256) void k_mad24(uint32_t* out, uint32_t a) {
  uint32_t x0 = threadIdx.x + a, x1 = x0 * 3, x2 = x0 * 5, x3 = x0 * 7;
  for (int i = 0; i < ITERS; i++) {
    asm volatile("v_mad_u32_u24 %0, %1, %2, %0\n\tv_mad_u32_u24 %1, %2, %3, %1\n\tv_mad_u32_u24 %2, %3, %0, %2\n\tv_mad_u32_u24 %3, %0, %1, %3"
                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
  }
  out[blockIdx.x * 256 + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3;
}
__global__ __launch_bounds__(256) void k_mulhi24(uint32_t* out, uint32_t a) {
  uint32_t x0 = threadIdx.x + a, x1 = x0 * 3, x2 = x0 * 5, x3 = x0 * 7;
  for (int i = 0; i < ITERS; i++) {
    asm volatile("v_mul_hi_u32_u24 %0, %1, %0\n\tv_mul_hi_u32_u24 %1, %2, %1\n\tv_mul_hi_u32_u24 %2, %3, %2\n\tv_mul_hi_u32_u24 %3, %0, %3"
                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
  }
  out[blockIdx.x * 256 + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3;
}
__global__ __launch_bounds__(256) void k_add3(uint32_t* out, uint32_t a) {
  uint32_t x0 = threadIdx.x + a, x1 = x0 * 3, x2 = x0 * 5, x3 = x0 * 7;
  for (int i = 0; i < ITERS; i++) {
    asm volatile("v_add3_u32 %0, %1, %2, %0\n\tv_add3_u32 %1, %2, %3, %1\n\tv_add3_u32 %2, %3, %0, %2\n\tv_add3_u32 %3, %0, %1, %3"
                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
  }
  out[blockIdx.x * 256 + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3;
}

// --- shipped field / curve arithmetic (fp.hpp, curve.hpp) ---------------------------------------------
// The saturated-32-bit-limb candidates this design was chosen against (C++ 96-bit accumulator 68 G/s, inline-asm
// v_mad_u64_u32 + v_addc 119 G/s, CIOS 95 G/s, vs 9 x 29-bit 173 G/s) were measured with an earlier revision of
// this file; their numbers are kept in profiles/r01_ubench_limb_formats.jsonl.
template <int FID> __global__ __launch_bounds__(256) void k_modmul(uint32_t* io, int iters) {
  int t = blockIdx.x * 256 + threadIdx.x;
  Fp<FID> x = Fp<FID>::from_words(io + 8 * t), y = Fp<FID>::from_words(io + 8 * t + 8);
  for (int i = 0; i < iters; i++) {
    x = x * y;
    y = y * x;
  }
  (x + y).norm().canon().to_words(io + 8 * t);
}
template <int FID> __global__ __launch_bounds__(256) void k_modsqr(uint32_t* io, int iters) {
  int t = blockIdx.x * 256 + threadIdx.x;
  Fp<FID> x = Fp<FID>::from_words(io + 8 * t), y = Fp<FID>::from_words(io + 8 * t + 8);
  for (int i = 0; i < iters; i++) {
    x = x.sqr();
    y = y.sqr();
  }
  (x + y).norm().canon().to_words(io + 8 * t);
}
// mixed addition throughput: every lane adds the same stream of 64 points to its own accumulator
template <int FID> __global__ __launch_bounds__(256) void k_madd(const AffineW* pts, XYZZW* out, int iters) {
  int t = blockIdx.x * 256 + threadIdx.x;
  XYZZ<FID> acc = XYZZ<FID>::from_affine(Affine<FID>::load(pts[t & 63]));
  for (int i = 0; i < iters; i++) acc.add_affine(Affine<FID>::load(pts[(t + i + 1) & 63]), (i & 1) != 0);
  acc.store(out[t]);
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

  ms = time_ms([&] { k_shr64_only<<<blocks, 256>>>((uint64_t*)buf, 3u); });
  printf("{\"ubench\": \"v_lshrrev_b64\", \"Gops\": %.1f}\n", lanes_ops / ms * 1e-6);
  ms = time_ms([&] { k_add64<<<blocks, 256>>>((uint64_t*)buf, 3u); });
  printf("{\"ubench\": \"v_lshl_add_u64\", \"Gops\": %.1f}\n", lanes_ops / ms * 1e-6);
  ms = time_ms([&] { k_alignbit<<<blocks, 256>>>((uint32_t*)buf, 3u); });
  printf("{\"ubench\": \"v_alignbit_b32\", \"Gops\": %.1f}\n", lanes_ops / ms * 1e-6);
  ms = time_ms([&] { k_mad24<<<blocks, 256>>>((uint32_t*)buf, 3u); });
  printf("{\"ubench\": \"v_mad_u32_u24\", \"Gops\": %.1f}\n", lanes_ops / ms * 1e-6);
  ms = time_ms([&] { k_mulhi24<<<blocks, 256>>>((uint32_t*)buf, 3u); });
  printf("{\"ubench\": \"v_mul_hi_u32_u24\", \"Gops\": %.1f}\n", lanes_ops / ms * 1e-6);
  ms = time_ms([&] { k_add3<<<blocks, 256>>>((uint32_t*)buf, 3u); });
  printf("{\"ubench\": \"v_add3_u32\", \"Gops\": %.1f}\n", lanes_ops / ms * 1e-6);

  // random-ish field elements < p: clear the top 3 bits
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
  const int it = 256;
  const double mm = (double)threads * it * 2;
  ms = time_ms([&] { k_modmul<0><<<blocks, 256>>>((uint32_t*)buf, it); });
  printf("{\"ubench\": \"modmul_9x29 bn254_fq\", \"Gmodmul_s\": %.2f}\n", mm / ms * 1e-6);
  ms = time_ms([&] { k_modsqr<0><<<blocks, 256>>>((uint32_t*)buf, it); });
  printf("{\"ubench\": \"modsqr_9x29 bn254_fq\", \"Gmodmul_s\": %.2f}\n", mm / ms * 1e-6);
  ms = time_ms([&] { k_modmul<2><<<blocks, 256>>>((uint32_t*)buf, it); });
  printf("{\"ubench\": \"modmul_9x29 pasta_fp\", \"Gmodmul_s\": %.2f}\n", mm / ms * 1e-6);
  // 64 points on BN254: (1,2) is on the curve; use doublings of it computed on the host via the same header
  {
    std::vector<AffineW> pts(64);
    uint32_t wx[8] = {1, 0, 0, 0, 0, 0, 0, 0}, wy[8] = {2, 0, 0, 0, 0, 0, 0, 0};
    Affine<0> g;
    g.x = Fp<0>::from_words(wx).to_internal().canon();
    g.y = Fp<0>::from_words(wy).to_internal().canon();
    XYZZ<0> acc = XYZZ<0>::from_affine(g);
    for (int i = 0; i < 64; i++) {
      acc.to_affine().store(pts[i]);
      acc.add_affine(g);
      acc.dbl_in_place();
    }
    AffineW* dp;
    XYZZW* dout;
    CHK(hipMalloc(&dp, 64 * sizeof(AffineW)));
    CHK(hipMalloc(&dout, (size_t)threads * sizeof(XYZZW)));
    CHK(hipMemcpy(dp, pts.data(), 64 * sizeof(AffineW), hipMemcpyHostToDevice));
    const int mit = 64;
    ms = time_ms([&] { k_madd<0><<<blocks, 256>>>(dp, dout, mit); });
    printf("{\"ubench\": \"xyzz_madd bn254 (L1-resident operands)\", \"Gmadd_s\": %.3f, \"blocks_per_cu\": 8}\n",
           (double)threads * mit / ms * 1e-6);
    ms = time_ms([&] { k_madd<0><<<blocks / 2, 256>>>(dp, dout, mit); });
    printf("{\"ubench\": \"xyzz_madd bn254 (L1-resident operands)\", \"Gmadd_s\": %.3f, \"blocks_per_cu\": 4}\n",
           (double)threads / 2 * mit / ms * 1e-6);
  }
  return 0;
}
