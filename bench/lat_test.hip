// bench/lat_test.hip -- latency of ONE dependent point addition in the forms the MSM tail uses (the floor of a reduction-tree
// level, which is two dependent additions): a chain of N additions acc += p_i run by a single wave (nothing else on the
// chip), for   (a) one lane per point (XYZZ::add, curve.hpp)   (b) four lanes per point (quad_add, curve_quad.hpp),
// each with the chained products (default) and the latency-oriented products (LAT).  Also the same chain with 8 waves per
// SIMD resident, which is the throughput side of the same code.
// Build: hipcc -O3 --offload-arch=gfx950 -I nova_amd/csrc -o bench/lat_test bench/lat_test.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../nova_amd/csrc/curve_quad.hpp"
using namespace nmx;
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <bool LAT> __global__ __launch_bounds__(256) void k_chain_lane(const XYZZW* pts, XYZZW* out, uint32_t n_add, uint32_t ring) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  XYZZ<0> acc = XYZZ<0>::load(pts[tid % ring]);
  for (uint32_t i = 0; i < n_add; i++) {
    XYZZ<0> p = XYZZ<0>::load(pts[(tid * 7u + i * 13u + 1u) % ring]);
    acc.template add<LAT>(p);
  }
  acc.store(out[tid]);
}
template <bool LAT> __global__ __launch_bounds__(256) void k_chain_quad(const XYZZW* pts, XYZZW* out, uint32_t n_add, uint32_t ring) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t q = tid & 3u, item = tid >> 2;
  Fp<0> acc = quad_load<0>(pts[item % ring], q);
  for (uint32_t i = 0; i < n_add; i++) {
    Fp<0> p = quad_load<0>(pts[(item * 7u + i * 13u + 1u) % ring], q);
    acc = quad_add<0, LAT>(acc, p, q);
  }
  quad_store<0>(out[item], q, acc);
}
template <class K> double run(K k, dim3 grid, dim3 block, const XYZZW* pts, XYZZW* out, uint32_t n_add, uint32_t ring) {
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  double best = 1e30;
  for (int r = 0; r < 5; r++) {
    CHK(hipEventRecord(e0));
    k<<<grid, block>>>(pts, out, n_add, ring);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  return best * 1e3;  // us
}
int main() {
  const uint32_t ring = 1021;
  std::vector<XYZZW> pts(ring);
  uint32_t wx[8] = {1, 0, 0, 0, 0, 0, 0, 0}, wy[8] = {2, 0, 0, 0, 0, 0, 0, 0};
  Affine<0> g; g.x = Fp<0>::from_words(wx).to_internal().canon(); g.y = Fp<0>::from_words(wy).to_internal().canon();
  XYZZ<0> acc = XYZZ<0>::from_affine(g);
  for (uint32_t i = 0; i < ring; i++) { acc.store(pts[i]); acc.add_affine(g); if (i % 3 == 0) acc.dbl_in_place(); }
  XYZZW *dp, *dout; CHK(hipMalloc(&dp, ring * sizeof(XYZZW))); CHK(hipMalloc(&dout, (1u << 20) * sizeof(XYZZW)));
  CHK(hipMemcpy(dp, pts.data(), ring * sizeof(XYZZW), hipMemcpyHostToDevice));
  const uint32_t N1 = 64, N2 = 576;
  auto per_add = [&](auto k, dim3 grid, dim3 block) {
    double a = run(k, grid, block, dp, dout, N1, ring), b = run(k, grid, block, dp, dout, N2, ring);
    return (b - a) / (N2 - N1);
  };
  printf("{\"what\": \"us per dependent full addition (XYZZ add-2008-s), BN254 Fq\",\n");
  printf(" \"one_wave\": {\"lane\": %.3f, \"lane_LAT\": %.3f, \"quad\": %.3f, \"quad_LAT\": %.3f},\n",
         per_add(k_chain_lane<false>, 1, 64), per_add(k_chain_lane<true>, 1, 64), per_add(k_chain_quad<false>, 1, 64), per_add(k_chain_quad<true>, 1, 64));
  printf(" \"one_wave_per_simd_all_cus\": {\"lane\": %.3f, \"quad\": %.3f},\n", per_add(k_chain_lane<false>, 256, 256), per_add(k_chain_quad<false>, 256, 256));
  printf(" \"eight_waves_per_simd\": {\"lane\": %.3f, \"lane_LAT\": %.3f, \"quad\": %.3f, \"quad_LAT\": %.3f}}\n",
         per_add(k_chain_lane<false>, 2048, 256), per_add(k_chain_lane<true>, 2048, 256), per_add(k_chain_quad<false>, 2048, 256), per_add(k_chain_quad<true>, 2048, 256));
  return 0;
}
