// bench/quad_test2.hip -- FoldQuadFn / ReducePairQuadFn vs FoldFn / ReducePairFn on the device (debug harness).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../nova_amd/csrc/curve_quad.hpp"
using namespace nmx;
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
template <class F> __global__ __launch_bounds__(256) void k_launch(F f, uint32_t n) {
  uint32_t tid = blockIdx.x * 256u + threadIdx.x;
  if (tid < n) f(tid);
}
template <class F> void launch(const F& f, uint32_t n) {
  k_launch<F><<<(n + 255) / 256, 256>>>(f, n);
  CHK(hipGetLastError());
  CHK(hipDeviceSynchronize());
}
static bool same(const XYZZW& a, const XYZZW& b) {
  Affine<0> x = XYZZ<0>::load(a).to_affine(), y = XYZZ<0>::load(b).to_affine();
  AffineW w1, w2; x.store(w1); y.store(w2);
  return memcmp(w1.w, w2.w, 64) == 0;
}
int main(int argc, char** argv) {
  int which = argc > 1 ? atoi(argv[1]) : 3;
  // points
  const int NP = 1024;
  std::vector<XYZZW> pts(NP);
  uint32_t wx[8] = {1, 0, 0, 0, 0, 0, 0, 0}, wy[8] = {2, 0, 0, 0, 0, 0, 0, 0};
  Affine<0> g; g.x = Fp<0>::from_words(wx).to_internal().canon(); g.y = Fp<0>::from_words(wy).to_internal().canon();
  XYZZ<0> acc = XYZZ<0>::from_affine(g);
  for (int i = 0; i < NP; i++) { acc.store(pts[i]); acc.add_affine(g); if (i % 3 == 0) acc.dbl_in_place(); }
  XYZZ<0>::identity().store(pts[5]);
  if (which & 1) {
    // fold: 3 heavy buckets with 5, 40, 300 partials
    std::vector<HeavyRec> heavy = {{0, 0, 5, 0}, {1, 5, 40, 0}, {2, 45, 300, 0}};
    uint32_t counters[4] = {345, 3, 0, 300};
    HeavyRec* dh; uint32_t* dc; XYZZW *p1, *p2, *b1, *b2;
    CHK(hipMalloc(&dh, sizeof(HeavyRec) * 3)); CHK(hipMalloc(&dc, 16));
    CHK(hipMalloc(&p1, 345 * 128)); CHK(hipMalloc(&p2, 345 * 128)); CHK(hipMalloc(&b1, 3 * 128)); CHK(hipMalloc(&b2, 3 * 128));
    CHK(hipMemcpy(dh, heavy.data(), sizeof(HeavyRec) * 3, hipMemcpyHostToDevice)); CHK(hipMemcpy(dc, counters, 16, hipMemcpyHostToDevice));
    CHK(hipMemcpy(p1, pts.data(), 345 * 128, hipMemcpyHostToDevice)); CHK(hipMemcpy(p2, pts.data(), 345 * 128, hipMemcpyHostToDevice));
    const uint32_t Ts[6] = {32768, 4096, 512, 64, 8, 1};
    for (int p = 0; p < 6; p++) {
      uint32_t T = Ts[p], cap = p == 0 ? 0xffffffffu : Ts[p - 1], groups = T == 1 ? 3 : 2;
      printf("fold pass T=%u\n", T); fflush(stdout);
      FoldFn<0> f{dc, dh, p1, b1, T, cap, groups};
      launch(f, groups * T);
      FoldQuadFn<0> fq{dc, dh, p2, b2, T, cap, groups};
      launch(fq, groups * T * 4);
    }
    XYZZW r1[3], r2[3];
    CHK(hipMemcpy(r1, b1, 384, hipMemcpyDeviceToHost)); CHK(hipMemcpy(r2, b2, 384, hipMemcpyDeviceToHost));
    for (int i = 0; i < 3; i++) printf("fold bucket %d: %s\n", i, same(r1[i], r2[i]) ? "same" : "DIFFERENT");
  }
  if (which & 2) {
    // reduce: 2 windows x 64 buckets
    const uint32_t W = 2, M = 64;
    XYZZW *B, *D1, *Y1, *D2, *Y2;
    CHK(hipMalloc(&B, W * M * 128)); CHK(hipMemcpy(B, pts.data() + 300, W * M * 128, hipMemcpyHostToDevice));
    const XYZZW *Da = B, *Ya = B, *Db = B, *Yb = B;
    uint32_t n_in = M, first = 1;
    while (n_in > 1) {
      uint32_t half = n_in / 2, pairs = W * half;
      printf("reduce level n_in=%u\n", n_in); fflush(stdout);
      CHK(hipMalloc(&D1, pairs * 128)); CHK(hipMalloc(&Y1, pairs * 128)); CHK(hipMalloc(&D2, pairs * 128)); CHK(hipMalloc(&Y2, pairs * 128));
      uint32_t pad = (pairs + 63u) & ~63u;
      ReducePairFn<0> f{Da, Ya, D1, Y1, n_in, pairs, pad, first};
      launch(f, 2 * pad);
      uint32_t padq = (pairs + 15u) & ~15u;
      ReducePairQuadFn<0> fq{Db, Yb, D2, Y2, n_in, pairs, padq, first};
      launch(fq, 2 * padq * 4);
      Da = D1; Ya = Y1; Db = D2; Yb = Y2; n_in = half; first = 0;
    }
    XYZZW r1[2], r2[2];
    CHK(hipMemcpy(r1, Ya, 256, hipMemcpyDeviceToHost)); CHK(hipMemcpy(r2, Yb, 256, hipMemcpyDeviceToHost));
    for (int i = 0; i < 2; i++) printf("reduce window %d: %s\n", i, same(r1[i], r2[i]) ? "same" : "DIFFERENT");
  }
  return 0;
}
