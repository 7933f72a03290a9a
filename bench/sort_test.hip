// sort_test.hip -- rocPRIM radix_sort_pairs: default onesweep (8 bits per pass) against wider digits, on the MSM's
// (bucket key, point index) shape.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 bench/sort_test.hip -o sort_test
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/rocprim.hpp>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <class Cfg> static int run(const char* name, uint32_t* k0, uint32_t* k1, uint32_t* v0, uint32_t* v1, size_t n,
                                    unsigned bits, const std::vector<uint32_t>& hk) {
  size_t tmpb = 0;
  CK((rocprim::radix_sort_pairs<Cfg>(nullptr, tmpb, k0, k1, v0, v1, n, 0u, bits, 0)));
  void* tmp;
  CK(hipMalloc(&tmp, tmpb));
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  float best = 1e9;
  for (int it = 0; it < 5; it++) {
    CK(hipEventRecord(a, 0));
    CK((rocprim::radix_sort_pairs<Cfg>(tmp, tmpb, k0, k1, v0, v1, n, 0u, bits, 0)));
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    best = std::min(best, ms);
  }
  std::vector<uint32_t> ok(n), ov(n);
  CK(hipMemcpy(ok.data(), k1, n * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(ov.data(), v1, n * 4, hipMemcpyDeviceToHost));
  bool good = true;
  for (size_t i = 0; i < n && good; i++) {
    if (i && ok[i - 1] > ok[i]) good = false;
    if (hk[ov[i]] != ok[i]) good = false;
  }
  printf("%-28s n=%zu bits=%u  %.3f ms  %.1f Gpairs/s  %s\n", name, n, bits, best, n / best / 1e6, good ? "sorted+stable-values-ok" : "WRONG");
  CK(hipFree(tmp));
  return 0;
}

int main(int argc, char** argv) {
  size_t n = argc > 1 ? strtoull(argv[1], 0, 10) : (size_t)13 << 22;
  unsigned bits = argc > 2 ? atoi(argv[2]) : 20;
  std::vector<uint32_t> hk(n), hv(n);
  uint64_t s = 88172645463325252ull;
  for (size_t i = 0; i < n; i++) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    hk[i] = (uint32_t)(s >> 20) & ((1u << bits) - 1);
    hv[i] = (uint32_t)i;
  }
  uint32_t *k0, *k1, *v0, *v1;
  CK(hipMalloc(&k0, n * 4)); CK(hipMalloc(&k1, n * 4)); CK(hipMalloc(&v0, n * 4)); CK(hipMalloc(&v1, n * 4));
  CK(hipMemcpy(k0, hk.data(), n * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(v0, hv.data(), n * 4, hipMemcpyHostToDevice));
  using namespace rocprim;
  if (run<default_config>("default", k0, k1, v0, v1, n, bits, hk)) return 1;
#ifdef NOMERGE
  using CN = radix_sort_config<default_config, default_config, default_config, NOMERGE>;
  if (run<CN>("onesweep below 1M (limit " "NOMERGE" ")", k0, k1, v0, v1, n, bits, hk)) return 1;
#endif
#ifdef WIDE
  using C10 = radix_sort_config<default_config, default_config,
                                radix_sort_onesweep_config<kernel_config<512, 32>, kernel_config<WIDE_BS, WIDE_IPT>, WIDE, block_radix_rank_algorithm::match>>;
  if (run<C10>("wide digits", k0, k1, v0, v1, n, bits, hk)) return 1;
#endif
  return 0;
}
