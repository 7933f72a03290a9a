// tests/cpp/sc_host_test.cpp -- g++-only harness around nova_amd/csrc/sc_host.hpp: the HOST side of the product's sum-check
// provers (round algebra + tail rounds) run as complete provers over host tables, behind the signatures of the oracle's
// ref_sumcheck_prove_* so that tests/test_sc_host.py can put both through the same checks without a GPU.  Test scaffolding: the
// product reaches this code only through nova_amd/csrc/sumcheck_prove.hpp.
#include <stdio.h>

#include "../../nova_amd/csrc/sc_host.hpp"

using namespace nmx;

namespace {
template <int FID> std::vector<HostFp4<FID>> load(const ScAlg<FID>& a, const uint8_t* v, size_t n) {
  std::vector<HostFp4<FID>> o(n);
  for (size_t i = 0; i < n; i++) o[i] = a.in(v + 32 * i);
  return o;
}
template <int FID, int MODE>
int prove(int mont, const uint8_t* claim, const uint8_t* taus, size_t nr, const uint8_t* A, const uint8_t* B, const uint8_t* C, TranscriptFn cb,
          void* ctx, uint8_t* out_polys, uint8_t* out_r, uint8_t* out_claims) {
  try {
    ScAlg<FID> alg(mont != 0);
    typename ScAlg<FID>::Eq eq;
    if (MODE == 3) eq.init(alg, taus, (uint32_t)nr);
    const size_t n = (size_t)1 << nr;
    auto a = load<FID>(alg, A, n), b = load<FID>(alg, B, n);
    std::vector<HostFp4<FID>> c;
    if (MODE == 3) c = load<FID>(alg, C, n);
    HostFp4<FID> cl = alg.in(claim);
    sc_tail_rounds<FID, MODE>(alg, &eq, (uint32_t)nr, 1, cl, a, b, c, cb, ctx, out_polys, out_r);
    alg.out(a[0], out_claims), alg.out(b[0], out_claims + 32);
    if (MODE == 3) alg.out(c[0], out_claims + 64);
    return 0;
  } catch (const ScFail& f) {
    fprintf(stderr, "sc_host_test: %s\n", f.msg.c_str());
    return -f.code;
  }
}
struct NoDevice {  // every polynomial is on the host from the start: the loop must never ask the device
  template <class H> H fail() const { throw ScFail{9, "device hook called in a host-only run"}; }
  void start(size_t) { throw ScFail{9, "device hook called in a host-only run"}; }
  template <int FID> HostFp4<FID> t(size_t) { throw ScFail{9, "device hook"}; }
};
template <int FID> struct NoDev {
  using H = HostFp4<FID>;
  void start(size_t) { throw ScFail{9, "device hook called in a host-only run"}; }
  H t0(size_t) { throw ScFail{9, "device hook called in a host-only run"}; }
  H t_m1(size_t) { throw ScFail{9, "device hook called in a host-only run"}; }
  void bind(size_t, const H&) { throw ScFail{9, "device hook called in a host-only run"}; }
  void ahead(size_t) { throw ScFail{9, "device hook called in a host-only run"}; }
};
template <int FID>
int batch(int mont, const uint8_t* claims, const size_t* num_rounds, const uint8_t* const* polys, const uint8_t* const* eq_points,
          const uint8_t* coeffs, size_t k, TranscriptFn cb, void* ctx, uint8_t* out_polys, uint8_t* out_r, uint8_t* out_finals) {
  try {
    ScAlg<FID> alg(mont != 0);
    std::vector<ScBatchClaim<FID>> cs(k);
    for (size_t i = 0; i < k; i++) {
      cs[i].num_rounds = (uint32_t)num_rounds[i];
      cs[i].eq.init(alg, eq_points[i], cs[i].num_rounds);
      cs[i].claim0 = cs[i].running = alg.in(claims + 32 * i);
      cs[i].coeff = alg.in(coeffs + 32 * i);
      cs[i].host = load<FID>(alg, polys[i], (size_t)1 << num_rounds[i]);
    }
    NoDev<FID> dev;
    sc_batch_rounds<FID>(alg, cs, dev, cb, ctx, out_polys, out_r, out_finals);
    return 0;
  } catch (const ScFail& f) {
    fprintf(stderr, "sc_host_test: %s\n", f.msg.c_str());
    return -f.code;
  }
}
}  // namespace

// UniPoly::from_evals_deg2 / _deg3 + evaluate as the product's host algebra spells them (ScAlg): evals = [f(0), f(1), leading coefficient(, f(-1))]
// canonical in, coefficients and the value at `at` canonical out -- for the reference's known answers (univariate.rs:284-355)
template <int FID> int unipoly(int deg, const uint8_t* evals, const uint8_t* at, uint8_t* out_coeffs, uint8_t* out_value) {
  using H = HostFp4<FID>;
  ScAlg<FID> alg(false);
  H e[4], co[4];
  for (int i = 0; i <= deg; i++) e[i] = alg.in(evals + 32 * i);
  const H claim = e[0] + e[1];  // the provers pass the round's claim f(0) + f(1), not f(1)
  if (deg == 2) ScAlg<FID>::from_evals_deg2(e[0], claim, e[2], co);
  else ScAlg<FID>::from_evals_deg3(e[0], claim, e[2], e[3], co);
  for (int i = 0; i <= deg; i++) alg.out(co[i], out_coeffs + 32 * i);
  alg.out(ScAlg<FID>::poly_eval(co, (uint32_t)deg + 1, alg.in(at)), out_value);
  return 0;
}
#define DISPATCH(call)        \
  switch (field) {            \
    case 0: return call(0);   \
    case 1: return call(1);   \
    case 2: return call(2);   \
    case 3: return call(3);   \
    default: return -100;     \
  }

extern "C" {
int hsc_unipoly_from_evals(int field, int deg, const uint8_t* evals, const uint8_t* at, uint8_t* out_coeffs, uint8_t* out_value) {
  if (deg != 2 && deg != 3) return -1;
#define CALL(F) (unipoly<F>(deg, evals, at, out_coeffs, out_value))
  DISPATCH(CALL)
#undef CALL
}
int hsc_prove_cubic3(int field, int mont, const uint8_t* claim, const uint8_t* taus, size_t nr, const uint8_t* A, const uint8_t* B,
                     const uint8_t* C, TranscriptFn cb, void* ctx, uint8_t* out_polys, uint8_t* out_r, uint8_t* out_claims) {
#define CALL(F) (prove<F, 3>(mont, claim, taus, nr, A, B, C, cb, ctx, out_polys, out_r, out_claims))
  DISPATCH(CALL)
#undef CALL
}
int hsc_prove_quad_prod(int field, int mont, const uint8_t* claim, size_t nr, const uint8_t* A, const uint8_t* B, TranscriptFn cb, void* ctx,
                        uint8_t* out_polys, uint8_t* out_r, uint8_t* out_claims) {
#define CALL(F) (prove<F, 4>(mont, claim, nullptr, nr, A, B, nullptr, cb, ctx, out_polys, out_r, out_claims))
  DISPATCH(CALL)
#undef CALL
}
int hsc_prove_batch_eval(int field, int mont, const uint8_t* claims, const size_t* num_rounds, const uint8_t* const* polys,
                         const uint8_t* const* eq_points, const uint8_t* coeffs, size_t k, TranscriptFn cb, void* ctx, uint8_t* out_polys,
                         uint8_t* out_r, uint8_t* out_finals) {
#define CALL(F) (batch<F>(mont, claims, num_rounds, polys, eq_points, coeffs, k, cb, ctx, out_polys, out_r, out_finals))
  DISPATCH(CALL)
#undef CALL
}
}
