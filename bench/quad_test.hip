// bench/quad_test.hip -- device check of the quad-cooperative add/dbl (curve_quad.hpp) against the single-lane code.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../nova_amd/csrc/curve_quad.hpp"
using namespace nmx;
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// out[3i] = A_i + B_i (quad), out[3i+1] = 2*A_i (quad), out[3i+2] = A_i + A_i (quad add -> doubling slow path)
__global__ void k_quad(const XYZZW* A, const XYZZW* B, XYZZW* out, int n) {
  uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t q = tid & 3, i = tid >> 2;
  if ((int)i >= n) return;
  Fp<0> a = quad_load<0>(A[i], q), b = quad_load<0>(B[i], q);
  quad_store<0>(out[3 * i], q, quad_add<0>(a, b, q));
  quad_store<0>(out[3 * i + 1], q, quad_dbl<0>(a, q));
  quad_store<0>(out[3 * i + 2], q, quad_add<0>(a, a, q));
}
__global__ void k_scalar(const XYZZW* A, const XYZZW* B, XYZZW* out, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  XYZZ<0> a = XYZZ<0>::load(A[i]), b = XYZZ<0>::load(B[i]);
  XYZZ<0> s = a; s.add(b); s.store(out[3 * i]);
  XYZZ<0> d = a; d.dbl_in_place(); d.store(out[3 * i + 1]);
  XYZZ<0> e = a; e.add(a); e.store(out[3 * i + 2]);
}
int main() {
  const int n = 256;
  std::vector<XYZZW> A(n), B(n);
  uint32_t wx[8] = {1, 0, 0, 0, 0, 0, 0, 0}, wy[8] = {2, 0, 0, 0, 0, 0, 0, 0};
  Affine<0> g; g.x = Fp<0>::from_words(wx).to_internal().canon(); g.y = Fp<0>::from_words(wy).to_internal().canon();
  XYZZ<0> acc = XYZZ<0>::from_affine(g), acc2 = acc; acc2.dbl_in_place(); acc2.add_affine(g);
  for (int i = 0; i < n; i++) {
    acc.store(A[i]); acc2.store(B[i]);           // projective (zz != 1) operands
    acc.add_affine(g); acc.dbl_in_place(); acc2.add(acc); acc2.add_affine(g);
  }
  XYZZ<0>::identity().store(A[7]); XYZZ<0>::identity().store(B[9]); B[11] = A[11];   // identity / equal operands
  { XYZZ<0> t = XYZZ<0>::load(A[13]).neg(); t.store(B[13]); }                         // P + (-P)
  XYZZW *dA, *dB, *o1, *o2;
  CHK(hipMalloc(&dA, n * 128)); CHK(hipMalloc(&dB, n * 128)); CHK(hipMalloc(&o1, 3 * n * 128)); CHK(hipMalloc(&o2, 3 * n * 128));
  CHK(hipMemcpy(dA, A.data(), n * 128, hipMemcpyHostToDevice)); CHK(hipMemcpy(dB, B.data(), n * 128, hipMemcpyHostToDevice));
  k_quad<<<(4 * n + 255) / 256, 256>>>(dA, dB, o1, n);
  CHK(hipDeviceSynchronize());
  k_scalar<<<(n + 255) / 256, 256>>>(dA, dB, o2, n);
  CHK(hipDeviceSynchronize());
  std::vector<XYZZW> r1(3 * n), r2(3 * n);
  CHK(hipMemcpy(r1.data(), o1, 3 * n * 128, hipMemcpyDeviceToHost)); CHK(hipMemcpy(r2.data(), o2, 3 * n * 128, hipMemcpyDeviceToHost));
  int bad = 0;
  for (int i = 0; i < 3 * n; i++) {
    // compare as affine points (representations may differ)
    Affine<0> a1 = XYZZ<0>::load(r1[i]).to_affine(), a2 = XYZZ<0>::load(r2[i]).to_affine();
    AffineW w1, w2; a1.store(w1); a2.store(w2);
    if (memcmp(w1.w, w2.w, 64) != 0) { if (bad < 5) printf("mismatch at %d (op %d)\n", i / 3, i % 3); bad++; }
  }
  printf("quad_test: %d mismatches of %d\n", bad, 3 * n);
  return bad != 0;
}
