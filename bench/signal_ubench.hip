// signal_ubench.hip -- how long does a word written by the HOST take to reach a waiting kernel?  Two places for the word:
//   pinned host memory (hipHostMallocCoherent): the kernel polls over PCIe;
//   device memory, fine-grained / uncached (hipExtMallocWithFlags), written by the CPU through the large BAR: the kernel polls HBM.
// And the reverse leg for reference: kernel writes pinned host memory, host polls.  Round trips of a ping-pong, 2000 iterations.
// Build: hipcc -O2 --offload-arch=gfx950 -o signal_ubench bench/signal_ubench.hip
#include <hip/hip_runtime.h>
#include <immintrin.h>
#include <stdio.h>
#include <chrono>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
// waits for *in == i (i = 1 .. n), answers by writing i to *out (pinned host memory)
__global__ void pingpong(const volatile u32x4* in, uint32_t* out, uint32_t n) {
  for (uint32_t i = 1; i <= n; i++) {
    uint64_t t0 = wall_clock64();
    for (;;) {
      u32x4 q = in[0];
      if (q.x == i) break;
      if (wall_clock64() - t0 > 100000000ull) return;  // 1 s
    }
    __hip_atomic_store(out, i, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
static double run(volatile uint32_t* host_in, const void* dev_in, bool wc, uint32_t* out_h, uint32_t* out_d, uint32_t n) {
  *out_h = 0;
  host_in[0] = 0;
  if (wc) _mm_sfence();
  hipLaunchKernelGGL(pingpong, dim3(1), dim3(1), 0, 0, (const volatile u32x4*)dev_in, out_d, n);
  auto t0 = std::chrono::steady_clock::now();
  for (uint32_t i = 1; i <= n; i++) {
    host_in[4] = i + 7;  // a payload word written before the flag, as the provers would
    if (wc) _mm_sfence();
    host_in[0] = i;
    if (wc) _mm_sfence();
    while (__atomic_load_n(out_h, __ATOMIC_ACQUIRE) != i) {
      if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(5)) return -1;
    }
  }
  double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n;
  (void)hipDeviceSynchronize();
  return us;
}
int main() {
  int largebar = 0;
  CHK(hipDeviceGetAttribute(&largebar, hipDeviceAttributeIsLargeBar, 0));
  printf("large BAR: %d\n", largebar);
  uint32_t *out_h, *out_d;
  CHK(hipHostMalloc((void**)&out_h, 256, hipHostMallocCoherent | hipHostMallocMapped));
  CHK(hipHostGetDevicePointer((void**)&out_d, out_h, 0));
  {
    uint32_t *h, *d;
    CHK(hipHostMalloc((void**)&h, 256, hipHostMallocCoherent | hipHostMallocMapped));
    CHK(hipHostGetDevicePointer((void**)&d, h, 0));
    for (int rep = 0; rep < 2; rep++) printf("flag in pinned host memory:            round trip %.2f us\n", run(h, d, false, out_h, out_d, 2000));
  }
  if (largebar) {
    for (int kind = 0; kind < 2; kind++) {
      uint32_t* d = nullptr;
      hipError_t e = hipExtMallocWithFlags((void**)&d, 4096, kind ? hipDeviceMallocUncached : hipDeviceMallocFinegrained);
      if (e != hipSuccess) { printf("hipExtMallocWithFlags(%d): %s\n", kind, hipGetErrorString(e)); continue; }
      CHK(hipMemset(d, 0, 4096));
      CHK(hipDeviceSynchronize());
      for (int rep = 0; rep < 2; rep++)
        printf("flag in %s device memory (CPU writes through the BAR): round trip %.2f us\n", kind ? "uncached" : "fine-grained", run(d, d, true, out_h, out_d, 2000));
    }
  }
  // launch latency for comparison: empty-ish kernel that writes the flag, host polls
  {
    auto t0 = std::chrono::steady_clock::now();
    uint32_t n = 500;
    uint32_t *h, *d;
    CHK(hipHostMalloc((void**)&h, 256, hipHostMallocCoherent | hipHostMallocMapped));
    CHK(hipHostGetDevicePointer((void**)&d, h, 0));
    for (uint32_t i = 1; i <= n; i++) {
      h[0] = i;
      *out_h = 0;
      hipLaunchKernelGGL(pingpong, dim3(1), dim3(1), 0, 0, (const volatile u32x4*)d, out_d, 1u);  // waits for 1: make it i == 1 always
      h[0] = 1;
      while (__atomic_load_n(out_h, __ATOMIC_ACQUIRE) != 1) {}
    }
    printf("launch -> flag back on the host:       %.2f us per launch\n", std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n);
    (void)hipDeviceSynchronize();
  }
  return 0;
}
