// nova_mi355x.hpp -- header-only C++ host mirror of the reference's provider interface for the commitment / MSM
// path, on top of the C ABI (nova_mi355x.h).  Same names, argument meaning and error behaviour as the reference:
//   DlogGroupExt            /root/reference/src/provider/traits.rs:77-117
//   CommitmentEngineTrait   /root/reference/src/traits/commitment.rs:52-195
//   Pedersen / HyperKZG CE  /root/reference/src/provider/pedersen.rs:240-305, src/provider/hyperkzg.rs:584-645
// No group arithmetic here: every operation is a call into libnova_mi355x.so.  Precondition violations that are
// `assert!` panics in the reference throw std::invalid_argument; library errors throw nova::provider::Error.
#pragma once
#include <array>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "nova_mi355x.h"

namespace nova {
namespace provider {

using Scalar = std::array<uint8_t, 32>;  // canonical little-endian (`to_repr()`), or raw Montgomery with mont = true
using Affine = std::array<uint8_t, 64>;  // x || y, identity = all zero (`to_coordinates()`, traits.rs:303-312)

struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string& m) : std::runtime_error("nmx error " + std::to_string(c) + ": " + m), code(c) {}
};
inline void check(int rc) {
  if (rc != NMX_OK) throw Error(rc, nmx_last_error());
}

// One host process, k GPUs (nmx_init_devices): keys of >= 2^20 points registered afterwards are sharded over the devices by
// the library; every MSM / commit below stays one synchronous call (the reference decomposes in-process too:
// /root/reference/src/provider/msm.rs:564-574).  Returns the number of devices in use.  count = 0: all visible GPUs.
inline int init_devices(int count = 0, bool oversubscribe = false) {
  check(nmx_init_devices(count, oversubscribe ? NMX_DEVICES_OVERSUBSCRIBE : 0u));
  return nmx_devices_in_use();
}

struct Point {  // result of an MSM / commit, as `to_coordinates()` returns it
  Affine xy{};
  bool is_inf = true;
  bool operator==(const Point& o) const { return xy == o.xy && is_inf == o.is_inf; }
};

// A commitment key resident in HBM (`CommitmentKey { ck, h }`, pedersen.rs:33-45 / hyperkzg.rs:84-100).
class CommitmentKey {
 public:
  CommitmentKey(int curve, const std::vector<Affine>& ck, const Affine& h, bool mont = false, bool precompute = true)
      : curve_(curve), n_(ck.size()), h_(h), mont_(mont) {
    uint32_t flags = (mont ? NMX_BASES_MONT : 0u) | (precompute ? NMX_BASES_PRECOMPUTE : 0u);
    check(nmx_bases_register(curve, ck.data(), ck.size(), flags, &handle_));
  }
  // `CommitmentEngineTrait::load_setup` (src/traits/commitment.rs:64-76).  HyperKZG (hyperkzg.rs:658-674): the first
  // n.next_power_of_two() tauG1 points of a .ptau file; `h` is derived from the label on the reference side and tau_H
  // (G2) stays with the host, so the caller supplies h.  Pedersen (pedersen.rs:318-340): "PEDERSEN_KEY" | h | ck.
  // The G1 points are validated as `read_points` does (ptau.rs:372-391: canonical coordinates, on the curve); errors throw
  // Error with NMX_E_IO / _FORMAT / _POINT.  The two G2 points of section 3 (tau_H) are NOT read or validated here: pairings
  // are outside this library, the caller that needs tau_H reads and checks it as the reference does.
  static CommitmentKey load_ptau(int curve, const std::string& path, size_t n, const Affine& h, bool precompute = true) {
    CommitmentKey k(curve, next_power_of_two(n), h);
    check(nmx_bases_register_ptau(curve, path.c_str(), k.n_, 2, precompute ? NMX_BASES_PRECOMPUTE : 0u, &k.handle_));
    return k;
  }
  static CommitmentKey load_keyfile(int curve, const std::string& path, size_t n, bool precompute = true) {
    CommitmentKey k(curve, next_power_of_two(n), Affine{});
    check(nmx_bases_register_keyfile(curve, path.c_str(), k.n_, precompute ? NMX_BASES_PRECOMPUTE : 0u, &k.handle_,
                                     k.h_.data()));
    return k;
  }
  // synthetic key P_i = (k0 + i) G built on the device, h = P_n (benchmarks and replays: nmx_bases_generate)
  static CommitmentKey generate(int curve, size_t n, uint64_t k0 = 1, bool precompute = true) {
    CommitmentKey k(curve, n, Affine{});
    check(nmx_bases_generate(curve, k0, n + 1, precompute ? NMX_BASES_PRECOMPUTE : 0u, &k.handle_));
    check(nmx_bases_read(k.handle_, n, 1, k.h_.data()));
    return k;
  }
  CommitmentKey(CommitmentKey&& o) noexcept : curve_(o.curve_), n_(o.n_), h_(o.h_), mont_(o.mont_), handle_(o.handle_) {
    o.handle_ = 0;
  }
  CommitmentKey(const CommitmentKey&) = delete;
  CommitmentKey& operator=(const CommitmentKey&) = delete;
  ~CommitmentKey() {
    if (handle_) nmx_bases_unregister(handle_);
  }
  size_t len() const { return n_; }
  int curve() const { return curve_; }
  uint64_t handle() const { return handle_; }
  const Affine& h() const { return h_; }
  bool mont() const { return mont_; }

 private:
  CommitmentKey(int curve, size_t n, const Affine& h) : curve_(curve), n_(n), h_(h), mont_(false) {}
  static size_t next_power_of_two(size_t n) {
    size_t p = 1;
    while (p < n) p <<= 1;
    return p;
  }
  int curve_;
  size_t n_;
  Affine h_;
  bool mont_;
  uint64_t handle_ = 0;
};

// `DlogGroupExt` for one curve id (NMX_BN254_G1 / NMX_GRUMPKIN / NMX_PALLAS / NMX_VESTA).
template <int CURVE> struct DlogGroupExt {
  // Smallest n worth sending to the GPU; below it the shim keeps the CPU msm() (pedersen.rs:484-497, msm.rs:233)
  static size_t min_gpu_n() { return nmx_min_gpu_n(CURVE); }
  // traits.rs:79 == msm() (msm.rs:225): assert_eq!(coeffs.len(), bases.len()).  Slice form: the library's slice cache
  // keeps `bases` resident (window tables included) keyed by bases.data(); a prefix of the same vector hits it.
  static Point vartime_multiscalar_mul(const std::vector<Scalar>& scalars, const std::vector<Affine>& bases,
                                       bool mont = false) {
    if (scalars.size() != bases.size()) throw std::invalid_argument("assert_eq!(coeffs.len(), bases.len())");
    Point p;
    uint8_t inf = 0;
    uint32_t flags = mont ? (NMX_SCALARS_MONT | NMX_BASES_MONT) : 0u;
    check(nmx_msm(CURVE, scalars.data(), bases.data(), scalars.size(), flags, p.xy.data(), &inf));
    p.is_inf = inf != 0;
    return p;
  }
  // same over a registered key prefix (`&ck.ck[..v.len()]`)
  static Point vartime_multiscalar_mul(const std::vector<Scalar>& scalars, const CommitmentKey& ck, bool mont = false) {
    if (scalars.size() > ck.len()) throw std::invalid_argument("assert!(ck.ck.len() >= v.len())");
    Point p;
    uint8_t inf = 0;
    check(nmx_msm_handle(ck.handle(), 0, scalars.data(), scalars.size(), mont ? NMX_SCALARS_MONT : 0u,
                         p.xy.data(), &inf));
    p.is_inf = inf != 0;
    return p;
  }
  // traits.rs:82-90: the j-th vector uses bases[..len_j]
  static std::vector<Point> batch_vartime_multiscalar_mul(const std::vector<std::vector<Scalar>>& scalars,
                                                          const CommitmentKey& ck, bool mont = false) {
    std::vector<const void*> ptrs;
    std::vector<size_t> lens;
    for (auto& v : scalars) {
      ptrs.push_back(v.data());
      lens.push_back(v.size());
    }
    std::vector<uint8_t> out(64 * scalars.size() + 1), inf(scalars.size() + 1);
    check(nmx_msm_batch_handle(ck.handle(), ptrs.data(), lens.data(), scalars.size(), mont ? NMX_SCALARS_MONT : 0u,
                               out.data(), inf.data()));
    std::vector<Point> r(scalars.size());
    for (size_t j = 0; j < scalars.size(); j++) {
      std::copy(out.begin() + 64 * j, out.begin() + 64 * j + 64, r[j].xy.begin());
      r[j].is_inf = inf[j] != 0;
    }
    return r;
  }
  // traits.rs:93-106 (`T: Into<u64>`: widen to u64 before the call)
  static Point vartime_multiscalar_mul_small(const std::vector<uint64_t>& scalars, const CommitmentKey& ck) {
    return vartime_multiscalar_mul_small_with_max_num_bits(scalars, ck, NMX_BITS_AUTO);
  }
  static Point vartime_multiscalar_mul_small_with_max_num_bits(const std::vector<uint64_t>& scalars,
                                                               const CommitmentKey& ck, uint32_t max_num_bits) {
    if (scalars.size() > ck.len()) throw std::invalid_argument("assert_eq!(bases.len(), scalars.len())");
    Point p;
    uint8_t inf = 0;
    check(nmx_msm_u64_handle(ck.handle(), 0, scalars.data(), scalars.size(), max_num_bits, 0, p.xy.data(), &inf));
    p.is_inf = inf != 0;
    return p;
  }
  // traits.rs:109-117: batch_vartime_multiscalar_mul_small -- the j-th vector uses bases[..len_j]
  static std::vector<Point> batch_vartime_multiscalar_mul_small(const std::vector<std::vector<uint64_t>>& scalars,
                                                                const CommitmentKey& ck, uint32_t max_num_bits = NMX_BITS_AUTO) {
    std::vector<const uint64_t*> ptrs;
    std::vector<size_t> lens;
    for (auto& v : scalars) {
      ptrs.push_back(v.data());
      lens.push_back(v.size());
    }
    std::vector<uint8_t> out(64 * scalars.size() + 1), inf(scalars.size() + 1);
    check(nmx_msm_u64_batch_handle(ck.handle(), ptrs.data(), lens.data(), scalars.size(), max_num_bits, 0u, out.data(), inf.data()));
    std::vector<Point> r(scalars.size());
    for (size_t j = 0; j < scalars.size(); j++) {
      std::copy(out.begin() + 64 * j, out.begin() + 64 * j + 64, r[j].xy.begin());
      r[j].is_inf = inf[j] != 0;
    }
    return r;
  }
};

// `CommitmentEngineTrait` restricted to the hot path.
template <int CURVE> struct CommitmentEngine {
  // pedersen.rs:263-270 / hyperkzg.rs:584-591:  msm(v, ck[..len v]) + h * r
  static Point commit(const CommitmentKey& ck, const std::vector<Scalar>& v, const Scalar& r, bool mont = false) {
    if (ck.len() < v.size()) throw std::invalid_argument("assert!(ck.ck.len() >= v.len())");
    Point p;
    uint8_t inf = 0;
    uint32_t flags = (mont ? NMX_SCALARS_MONT : 0u) | (ck.mont() ? NMX_BASES_MONT : 0u);
    check(nmx_commit(ck.handle(), v.data(), v.size(), ck.h().data(), r.data(), flags, p.xy.data(), &inf));
    p.is_inf = inf != 0;
    return p;
  }
  // The same commitment begun now and collected later (nmx_commit_begin / nmx_commit_finish): `W.commit(ck)` beside the cross
  // term + commit(T) that does not read it (src/r1cs/mod.rs:590-622, src/nova/nifs.rs:53-63).  `v` must outlive finish().
  class Pending {
    uint64_t ticket_ = 0;

   public:
    explicit Pending(uint64_t t) : ticket_(t) {}
    Pending(Pending&& o) noexcept : ticket_(o.ticket_) { o.ticket_ = 0; }
    Pending(const Pending&) = delete;
    Pending& operator=(const Pending&) = delete;
    ~Pending() {  // never leave a ticket behind: wait, drop the result
      uint8_t xy[128], inf;
      if (ticket_) (void)nmx_commit_finish(ticket_, xy, &inf);
    }
    Point finish() {
      Point p;
      uint8_t inf = 0;
      const uint64_t t = ticket_;
      ticket_ = 0;
      check(nmx_commit_finish(t, p.xy.data(), &inf));
      p.is_inf = inf != 0;
      return p;
    }
  };
  static Pending commit_begin(const CommitmentKey& ck, const std::vector<Scalar>& v, const Scalar& r, bool mont = false) {
    if (ck.len() < v.size()) throw std::invalid_argument("assert!(ck.ck.len() >= v.len())");
    uint64_t t = 0;
    uint32_t flags = (mont ? NMX_SCALARS_MONT : 0u) | (ck.mont() ? NMX_BASES_MONT : 0u);
    check(nmx_commit_begin(ck.handle(), v.data(), v.size(), ck.h().data(), r.data(), flags, &t));
    return Pending(t);
  }
  // hyperkzg.rs:593-612 with r_i = 0 (the HyperKZG prover's use)
  static std::vector<Point> batch_commit(const CommitmentKey& ck, const std::vector<std::vector<Scalar>>& v) {
    return DlogGroupExt<CURVE>::batch_vartime_multiscalar_mul(v, ck);
  }
};

}  // namespace provider

// Spartan's sum-check provers (src/spartan/sumcheck.rs) over the C ABI's one-call-per-prover entry points.  `Transcript` is the
// caller's: any type with `Scalar round(const std::vector<Scalar>& unipoly_coeffs)` -- what the reference writes as
// `transcript.absorb(b"p", &poly); transcript.squeeze(b"c")` (sumcheck.rs:224-227, 481-484, 315-318).  Tables are host vectors here
// (uploaded for the call; hosts that keep their vectors in HBM call the C entry points with NMX_SCALARS_DEVICE).
namespace spartan {
using provider::Scalar;
using provider::check;

struct SumcheckProof {                              // what SumcheckProof::new(compressed_polys) is built from, plus r and the claims
  std::vector<std::vector<Scalar>> polys;           // per round: UniPoly coefficients, constant term first
  std::vector<Scalar> r;                            // the challenges
  std::vector<Scalar> claims;                       // final evaluations (poly_A[0], ...)
};
namespace detail {
template <class Transcript> int round_cb(void* ctx, const uint8_t* coeffs32, size_t n, uint8_t* out) {
  try {
    std::vector<Scalar> co(n);
    for (size_t i = 0; i < n; i++) std::copy(coeffs32 + 32 * i, coeffs32 + 32 * i + 32, co[i].begin());
    const Scalar r = static_cast<Transcript*>(ctx)->round(co);
    std::copy(r.begin(), r.end(), out);
    return 0;
  } catch (...) {
    return 1;  // never let an exception cross the C frame: the call fails with NMX_E_ARG
  }
}
inline SumcheckProof unpack(size_t rounds, size_t nco, size_t nclaims, const std::vector<uint8_t>& p, const std::vector<uint8_t>& r,
                            const std::vector<uint8_t>& c) {
  SumcheckProof out;
  out.polys.assign(rounds, std::vector<Scalar>(nco));
  out.r.resize(rounds), out.claims.resize(nclaims);
  for (size_t j = 0; j < rounds; j++) {
    for (size_t i = 0; i < nco; i++) std::copy(p.begin() + 32 * (nco * j + i), p.begin() + 32 * (nco * j + i) + 32, out.polys[j][i].begin());
    std::copy(r.begin() + 32 * j, r.begin() + 32 * j + 32, out.r[j].begin());
  }
  for (size_t i = 0; i < nclaims; i++) std::copy(c.begin() + 32 * i, c.begin() + 32 * i + 32, out.claims[i].begin());
  return out;
}
}  // namespace detail

template <int FIELD> struct Sumcheck {
  // SumcheckProof::prove_cubic_with_three_inputs (sumcheck.rs:446-507)
  template <class Transcript>
  static SumcheckProof prove_cubic_with_three_inputs(const Scalar& claim, const std::vector<Scalar>& taus, const std::vector<Scalar>& A,
                                                     const std::vector<Scalar>& B, const std::vector<Scalar>& C, Transcript& tr, bool mont = false) {
    const size_t l = taus.size();
    if (A.size() != (size_t)1 << l || B.size() != A.size() || C.size() != A.size()) throw std::invalid_argument("tables must hold 2^taus.len() elements");
    std::vector<uint8_t> p(128 * l + 1), r(32 * l + 1), c(96);
    check(nmx_sumcheck_prove_cubic_with_three_inputs(FIELD, claim.data(), taus.data(), l, const_cast<Scalar*>(A.data()), const_cast<Scalar*>(B.data()),
                                                     const_cast<Scalar*>(C.data()), mont ? NMX_SCALARS_MONT : 0u, &detail::round_cb<Transcript>, &tr,
                                                     p.data(), r.data(), c.data()));
    return detail::unpack(l, 4, 3, p, r, c);
  }
  // SumcheckProof::prove_quad_prod (sumcheck.rs:199-249)
  template <class Transcript>
  static SumcheckProof prove_quad_prod(const Scalar& claim, size_t num_rounds, const std::vector<Scalar>& A, const std::vector<Scalar>& B,
                                       Transcript& tr, bool mont = false) {
    if (A.size() != (size_t)1 << num_rounds || B.size() != A.size()) throw std::invalid_argument("tables must hold 2^num_rounds elements");
    std::vector<uint8_t> p(96 * num_rounds + 1), r(32 * num_rounds + 1), c(64);
    check(nmx_sumcheck_prove_quad_prod(FIELD, claim.data(), num_rounds, const_cast<Scalar*>(A.data()), const_cast<Scalar*>(B.data()),
                                       mont ? NMX_SCALARS_MONT : 0u, &detail::round_cb<Transcript>, &tr, p.data(), r.data(), c.data()));
    return detail::unpack(num_rounds, 3, 2, p, r, c);
  }
  // SumcheckProof::prove_batch_eval (sumcheck.rs:251-353)
  template <class Transcript>
  static SumcheckProof prove_batch_eval(const std::vector<Scalar>& claims, const std::vector<std::vector<Scalar>>& polys,
                                        const std::vector<std::vector<Scalar>>& eq_points, const std::vector<Scalar>& coeffs, Transcript& tr,
                                        bool mont = false) {
    const size_t k = polys.size();
    if (claims.size() != k || eq_points.size() != k || coeffs.size() != k || k == 0) throw std::invalid_argument("one claim, point and coefficient per polynomial");
    std::vector<size_t> nr(k);
    std::vector<void*> pp(k);
    std::vector<const void*> qp(k);
    size_t nmax = 0;
    for (size_t i = 0; i < k; i++) {
      nr[i] = eq_points[i].size();
      if (polys[i].size() != (size_t)1 << nr[i]) throw std::invalid_argument("poly size mismatch");
      pp[i] = const_cast<Scalar*>(polys[i].data()), qp[i] = eq_points[i].data();
      nmax = nr[i] > nmax ? nr[i] : nmax;
    }
    std::vector<uint8_t> p(96 * nmax + 1), r(32 * nmax + 1), f(32 * k);
    check(nmx_sumcheck_prove_batch_eval(FIELD, claims.data(), nr.data(), pp.data(), qp.data(), coeffs.data(), k, mont ? NMX_SCALARS_MONT : 0u,
                                        &detail::round_cb<Transcript>, &tr, p.data(), r.data(), f.data()));
    return detail::unpack(nmax, 3, k, p, r, f);
  }
};
}  // namespace spartan

// The inner-product argument (src/provider/ipa_pc.rs:174-281) over the C ABI's one call.  `Transcript` is the caller's: any type with
// `Scalar round(const Point& L, const Point& R)` -- `transcript.absorb(b"L", &L); transcript.absorb(b"R", &R); transcript.squeeze(b"r")`
// (:231-234).  Host vectors here (uploaded for the call); `ck_c` is the already scaled one-point key (`ck_c.scale(&r)`, :190-191).
namespace ipa {
using provider::Affine;
using provider::check;
using provider::CommitmentKey;
using provider::Point;
using provider::Scalar;

struct InnerProductArgument {  // L_vec, R_vec, a_hat (:157-162)
  std::vector<Point> L_vec, R_vec;
  Scalar a_hat{};
};
namespace detail {
template <class Transcript> int round_cb(void* ctx, const uint8_t* L, int L_inf, const uint8_t* R, int R_inf, uint8_t* out) {
  try {
    Point l, r;
    std::copy(L, L + 64, l.xy.begin()), std::copy(R, R + 64, r.xy.begin());
    l.is_inf = L_inf != 0, r.is_inf = R_inf != 0;
    const Scalar c = static_cast<Transcript*>(ctx)->round(l, r);
    std::copy(c.begin(), c.end(), out);
    return 0;
  } catch (...) {
    return 1;  // never let an exception cross the C frame: the call fails with NMX_E_ARG
  }
}
}  // namespace detail
template <class Transcript>
InnerProductArgument prove(const CommitmentKey& ck, const Affine& ck_c, const std::vector<Scalar>& a_vec, const std::vector<Scalar>& b_vec,
                           Transcript& tr, bool mont = false) {
  if (a_vec.size() != b_vec.size() || a_vec.empty()) throw std::invalid_argument("InvalidInputLength (ipa_pc.rs:185-187)");
  const size_t n = a_vec.size();
  size_t rounds = 0;
  while (((size_t)1 << rounds) < n) rounds++;
  std::vector<uint8_t> L(64 * rounds + 1), R(64 * rounds + 1), inf(2 * rounds + 1);
  InnerProductArgument out;
  check(nmx_ipa_prove(ck.handle(), ck_c.data(), a_vec.data(), b_vec.data(), n, mont ? NMX_SCALARS_MONT : 0u, &detail::round_cb<Transcript>, &tr,
                      L.data(), R.data(), inf.data(), out.a_hat.data()));
  out.L_vec.resize(rounds), out.R_vec.resize(rounds);
  for (size_t k = 0; k < rounds; k++) {
    std::copy(L.begin() + 64 * k, L.begin() + 64 * k + 64, out.L_vec[k].xy.begin());
    std::copy(R.begin() + 64 * k, R.begin() + 64 * k + 64, out.R_vec[k].xy.begin());
    out.L_vec[k].is_inf = inf[2 * k] != 0, out.R_vec[k].is_inf = inf[2 * k + 1] != 0;
  }
  return out;
}
}  // namespace ipa

// Hosts that keep their vectors in HBM between provider calls (INTEGRATION.md sections 2b-2e: the patched r1cs/mod.rs, nifs.rs,
// snark.rs): the same operations over device pointers.  Thin by design -- each function is one C call with NMX_SCALARS_DEVICE (and
// NMX_ASYNC where the host does not look at the result before its next synchronous call); allocation is the host's business
// (hipMalloc, a pool, torch).  `n` counts 32-byte elements.  Used by bench/csnark_replay.cpp (the chained CompressedSNARK::prove replay).
namespace resident {
using provider::Affine;
using provider::check;
using provider::CommitmentKey;
using provider::Point;
using provider::Scalar;
constexpr uint32_t kDev = NMX_SCALARS_DEVICE, kAsync = NMX_SCALARS_DEVICE | NMX_ASYNC;

inline Point commit(const CommitmentKey& ck, const void* v, size_t n, const Scalar& r) {  // CE::commit (pedersen.rs:263-270, hyperkzg.rs:584-591)
  Point p;
  uint8_t inf = 0;
  check(nmx_commit(ck.handle(), v, n, ck.h().data(), r.data(), kDev, p.xy.data(), &inf));
  p.is_inf = inf != 0;
  return p;
}
inline uint64_t commit_begin(const CommitmentKey& ck, const void* v, size_t n, const Scalar& r) {  // one arm of a rayon::join of two commits
  uint64_t t = 0;
  check(nmx_commit_begin(ck.handle(), v, n, ck.h().data(), r.data(), kDev, &t));
  return t;
}
inline Point commit_finish(uint64_t ticket) {
  Point p;
  uint8_t inf = 0;
  check(nmx_commit_finish(ticket, p.xy.data(), &inf));
  p.is_inf = inf != 0;
  return p;
}
inline std::vector<Point> batch_commit(const CommitmentKey& ck, const std::vector<const void*>& vs, const std::vector<size_t>& lens) {  // hyperkzg.rs:593-612, r_i = 0
  const size_t k = vs.size();
  std::vector<uint8_t> xy(64 * (k ? k : 1)), inf(k ? k : 1);
  check(nmx_msm_batch_handle(ck.handle(), vs.data(), lens.data(), k, kDev, xy.data(), inf.data()));
  std::vector<Point> out(k);
  for (size_t i = 0; i < k; i++) {
    std::copy(xy.begin() + 64 * i, xy.begin() + 64 * i + 64, out[i].xy.begin());
    out[i].is_inf = inf[i] != 0;
  }
  return out;
}
// z = [W | u | X | 0 ...] (snark.rs:133, r1cs/mod.rs:638-639)
inline void concat_z(int field, const void* W, size_t n, const Scalar& u, const Scalar& X, size_t n_out, void* out) {
  const void* parts[3] = {W, u.data(), X.data()};
  const size_t lens[3] = {n, 1, 1};
  check(nmx_field_concat(field, parts, lens, 1u, 3, n_out, kAsync, out));
}
inline void clone(int field, const void* v, size_t n, void* out) {
  const void* parts[1] = {v};
  const size_t lens[1] = {n};
  check(nmx_field_concat(field, parts, lens, 1u, 1, n, kAsync, out));
}
inline void vec_add(int field, const void* a, const void* b, size_t n, void* out) { check(nmx_field_vec_add(field, a, b, n, kAsync, out)); }
inline void axpy(int field, const void* a, const void* b, const Scalar& r, size_t n, void* out) { check(nmx_field_axpy(field, a, b, r.data(), n, kAsync, out)); }
inline void axpy2(int field, const void* a, const void* b, const void* c, const Scalar& r, size_t n, void* out) {
  check(nmx_field_axpy2(field, a, b, c, r.data(), n, kAsync, out));
}
inline void cross_term2(int field, const void* az, const void* bz, const void* cz, const void* e1, const void* e2, const Scalar& u, size_t n, void* out) {
  check(nmx_field_cross_term2(field, az, bz, cz, e1, e2, u.data(), n, kAsync, out));
}
// R1CSShape::multiply_vec / compute_eval_table_sparse: the three products in one call
inline void multiply_vec3(const uint64_t mats[3], bool transposed, const void* x, size_t x_len, void* const outs[3]) {
  check(nmx_spmv_apply_many(mats, 3, transposed ? 1 : 0, x, x_len, kAsync, outs));
}
// AZ o BZ - u CZ - E over Z = z (sample_random_instance_witness with E = 0, r1cs/mod.rs:803-812)
inline void r1cs_cross_term(const uint64_t mats[3], const void* z, size_t z_len, const void* e, const Scalar& u, void* out) {
  check(nmx_r1cs_cross_term(mats[0], mats[1], mats[2], z, nullptr, z_len, e, u.data(), kAsync, out));
}
inline Scalar mle_evaluate(int field, const void* z, size_t len, const std::vector<Scalar>& r) {
  Scalar out;
  check(nmx_mle_evaluate(field, z, len, r.data(), r.size(), kDev, out.data()));
  return out;
}
inline std::vector<Scalar> mle_multi_evaluate(int field, const std::vector<const void*>& zs, size_t len, const std::vector<Scalar>& r) {
  std::vector<Scalar> out(zs.size());
  check(nmx_mle_multi_evaluate(field, zs.data(), zs.size(), len, r.data(), r.size(), kDev, out[0].data()));
  return out;
}
inline void eq_evals(int field, const std::vector<Scalar>& r, void* out) { check(nmx_eq_evals_from_points(field, r.data(), r.size(), kDev, out)); }
inline void lincomb_powers(int field, const std::vector<const void*>& vs, const std::vector<size_t>& lens, const Scalar& s, size_t n_out, void* out) {
  check(nmx_field_lincomb_powers(field, vs.data(), lens.data(), vs.size(), s.data(), n_out, kDev, out));
}
inline void fold_pairs(int field, const void* p, size_t len, const Scalar& x, void* out) { check(nmx_poly_fold_pairs(field, p, len, x.data(), kAsync, out)); }
// the fold loop of hyperkzg.rs:1085-1095 in one call: outs[i] = fold(outs[i - 1] or p, xs[i])
inline void fold_chain(int field, const void* p, size_t len, const std::vector<Scalar>& xs, const std::vector<void*>& outs) {
  check(nmx_poly_fold_chain(field, p, len, xs.data(), xs.size(), kAsync, outs.data()));
}
inline void suffix_horner(int field, const void* f, size_t n, const Scalar& u, void* out) { check(nmx_poly_suffix_horner(field, f, n, u.data(), kDev, out)); }
inline std::vector<Scalar> poly_eval_multi(int field, const std::vector<const void*>& polys, const std::vector<size_t>& lens, const std::vector<Scalar>& pts) {
  std::vector<Scalar> out(polys.size() * pts.size());
  check(nmx_poly_eval_multi(field, polys.data(), lens.data(), polys.size(), pts.data(), pts.size(), kDev, out[0].data()));
  return out;
}
// the sum-check provers over HBM-resident tables (bound in place); `cb` / `ctx`: nmx_transcript_fn and its state
struct Proof {
  std::vector<uint8_t> polys, r, claims;
};
inline Proof prove_cubic(int field, const Scalar& claim, const std::vector<Scalar>& taus, void* A, void* B, void* C, nmx_transcript_fn cb, void* ctx) {
  const size_t l = taus.size();
  Proof p{std::vector<uint8_t>(128 * l), std::vector<uint8_t>(32 * l), std::vector<uint8_t>(96)};
  check(nmx_sumcheck_prove_cubic_with_three_inputs(field, claim.data(), taus.data(), l, A, B, C, kDev, cb, ctx, p.polys.data(), p.r.data(), p.claims.data()));
  return p;
}
inline Proof prove_quad(int field, const Scalar& claim, size_t l, void* A, void* B, nmx_transcript_fn cb, void* ctx) {
  Proof p{std::vector<uint8_t>(96 * l), std::vector<uint8_t>(32 * l), std::vector<uint8_t>(64)};
  check(nmx_sumcheck_prove_quad_prod(field, claim.data(), l, A, B, kDev, cb, ctx, p.polys.data(), p.r.data(), p.claims.data()));
  return p;
}
inline Proof prove_batch(int field, const std::vector<Scalar>& claims, const std::vector<size_t>& nr, const std::vector<void*>& polys,
                         const std::vector<const void*>& points, const std::vector<Scalar>& coeffs, nmx_transcript_fn cb, void* ctx) {
  size_t nmax = 0;
  for (size_t v : nr) nmax = v > nmax ? v : nmax;
  Proof p{std::vector<uint8_t>(96 * nmax), std::vector<uint8_t>(32 * nmax), std::vector<uint8_t>(32 * claims.size())};
  check(nmx_sumcheck_prove_batch_eval(field, claims.data(), nr.data(), polys.data(), points.data(), coeffs.data(), claims.size(), kDev, cb, ctx,
                                      p.polys.data(), p.r.data(), p.claims.data()));
  return p;
}
// ck_c.scale(&r) (src/provider/ipa_pc.rs:190-191, pedersen.rs:499-506): one point times one scalar -- a commitment to the empty
// vector with h = the point
inline Affine scale_point(const CommitmentKey& ck, const Affine& h, const Scalar& r) {
  Affine out{};
  uint8_t inf = 0;
  check(nmx_commit(ck.handle(), nullptr, 0, h.data(), r.data(), 0, out.data(), &inf));
  return out;
}
// InnerProductArgument::prove (src/provider/ipa_pc.rs:174-281) over HBM-resident a, b; `ck_c` already scaled (scale_point)
struct IpaProof {
  std::vector<uint8_t> L, R, inf;  // log2 n points of 64 bytes each; inf[2k] / inf[2k + 1]: L_k / R_k is the identity
  Scalar a_hat{};
};
inline IpaProof ipa_prove(const CommitmentKey& ck, const Affine& ck_c, const void* a, const void* b, size_t n, nmx_ipa_transcript_fn cb, void* ctx) {
  size_t rounds = 0;
  while (((size_t)1 << rounds) < n) rounds++;
  IpaProof p{std::vector<uint8_t>(64 * (rounds ? rounds : 1)), std::vector<uint8_t>(64 * (rounds ? rounds : 1)), std::vector<uint8_t>(2 * (rounds ? rounds : 1)), {}};
  check(nmx_ipa_prove(ck.handle(), ck_c.data(), a, b, n, kDev, cb, ctx, p.L.data(), p.R.data(), p.inf.data(), p.a_hat.data()));
  p.L.resize(64 * rounds), p.R.resize(64 * rounds), p.inf.resize(2 * rounds);
  return p;
}
}  // namespace resident
}  // namespace nova
