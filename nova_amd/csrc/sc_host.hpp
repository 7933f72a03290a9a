// sc_host.hpp -- HOST side of the sum-check provers (sumcheck_prove.hpp): the O(1) algebra of a round and the TAIL rounds, in
// HostFp4 arithmetic.  Host-only and free of HIP, so that tests/cpp/sc_host_test.cpp can run it under g++ against the oracle
// on a machine without a GPU (tests/test_sc_host.py): with the tail threshold above the instance size the tail IS a complete
// prover.
//
// Reference (paths under /root/reference/src/spartan):
//   EqSumCheckInstance::new / derive_from_claim_deg2 / _deg1 / bound         sumcheck.rs:608-677, 680-753, 1226-1231
//   evaluation_points_cubic_with_three_inputs / quadratic_with_one_input     sumcheck.rs:900-970, 1039-1083
//   fallback_eval_inf_three_inputs / _one_input (tau = 0)                    sumcheck.rs:1085-1136, 1185-1222
//   compute_eval_points_quad_prod                                            sumcheck.rs:163-186
//   UniPoly::from_evals_deg2 / _deg3 / evaluate                              polys/univariate.rs:90-113, 140-149
//   SumcheckProof::update_claim                                              sumcheck.rs:68-75
//   MultilinearPolynomial::bind_poly_var_top                                 polys/multilinear.rs:65-84
// Why a tail on the host at all: a round whose tables hold <= 64 elements is ~300 field products; a kernel round trip (launch,
// one block, result through PCIe) is 20-25 us whatever the size, the host does the same round in 1-8 us.  The tables reach the
// host once (the last device bind writes them into pinned memory) and the remaining log2(64) = 6 rounds never touch the GPU.
#pragma once
#include <stddef.h>

#include <string>
#include <vector>

#include "host_fp4.hpp"

namespace nmx {

// the transcript's side of a round (include/nova_mi355x.h nmx_transcript_fn)
using TranscriptFn = int (*)(void* ctx, const uint8_t* coeffs, size_t n_coeffs, uint8_t* challenge32);
struct ScFail {  // carried out to the C boundary by the caller (Fail in the library, an error code in the g++ test)
  int code;      // 1: transcript callback failed, 2: challenge >= p
  std::string msg;
};

template <int FID> struct ScAlg {
  using H = HostFp4<FID>;
  bool mont;  // the vectors' own form: halo2curves Montgomery words (true) or canonical (false)
  explicit ScAlg(bool mont_) : mont(mont_) {}

  H in(const void* p) const {
    uint32_t w[8];
    memcpy(w, p, 32);
    if (!Fp<FID>::words_lt_p(w)) throw ScFail{2, "challenge / claim >= field modulus"};
    return mont ? H::from_mont256(p) : H::from_canonical(p);
  }
  void out(const H& v, uint8_t* p) const {
    if (mont) v.to_mont256(p);
    else v.to_canonical(p);
  }
  static H two_inv() {
    static const H t = H::from_u64(2).inv();
    return t;
  }
  // transcript step: coefficients out (the vectors' own form), challenge in
  H ask(TranscriptFn cb, void* ctx, const H* coeffs, uint32_t n, uint8_t* polys_out, uint8_t* r_out, double* cb_ms = nullptr) const;
  // UniPoly::evaluate (univariate.rs:140-149)
  static H poly_eval(const H* co, uint32_t n, const H& r) {
    H eval = co[0], power = r;
    for (uint32_t i = 1; i < n; i++) {
      eval = eval + power * co[i];
      power = power * r;
    }
    return eval;
  }
  // UniPoly::from_evals_deg3 on [s(0), claim - s(0), cubic coefficient, s(-1)] (univariate.rs:103-113)
  static void from_evals_deg3(const H& s0, const H& claim, const H& lead, const H& sm1, H co[4]) {
    const H s1 = claim - s0;
    co[0] = s0, co[3] = lead;
    co[2] = (s1 + sm1) * two_inv() - s0;
    co[1] = s1 - lead - s0 - co[2];
  }
  // UniPoly::from_evals_deg2 on [e0, claim - e0, quadratic coefficient] (univariate.rs:90-99)
  static void from_evals_deg2(const H& e0, const H& claim, const H& quad, H co[3]) {
    const H s1 = claim - e0;
    co[0] = e0, co[2] = quad;
    co[1] = s1 - quad - e0;
  }
  // SumcheckProof::update_claim (sumcheck.rs:68-75) with evals [e0, 0, em1]
  static H update_claim(const H& claim, const H& e0, const H& em1, const H& r) {
    const H e1 = claim - e0;
    const H a1 = (e1 - em1) * two_inv(), a2 = (e1 + em1) * two_inv() - e0;
    return e0 + r * (a1 + r * a2);
  }
  static H pow2(uint32_t e) { return H::pow2(e); }

  // EqSumCheckInstance: the scalars (the sqrt-size tables live on the device; the tail builds the small ones it needs here)
  struct Eq {
    uint32_t l = 0, first_half = 0, second_half = 0, round = 1;
    std::vector<H> taus, eq0, slope, eqm1;  // eq_tau_0_a_inf (sumcheck.rs:643-652): eq(tau, 0), 2 tau - 1, eq(tau, -1)
    H eval_eq_left;
    // derive_from_claim divides by l(1) p = tau_j x eval_eq_left once per round (sumcheck.rs:680-753).  Rounds 1-5 of this project did
    // that inversion on the host every round (~3 us each, prepared under the device's pass where there was one: at small tables and in
    // the batch prover -- two claims, short passes -- it was on the critical path: ~120 us of a 2^20 batch proof).  Round 6 tracks the
    // INNER claim instead: the round polynomial is s_j(X) = p_j eq(tau_j, X) t_j(X) with p_j = eval_eq_left, and the claim of round j
    // is s_{j-1}(r_{j-1}) = p_j t_{j-1}(r_{j-1}); so with T_j := t_{j-1}(r_{j-1}) (T_1 = the caller's claim, p_1 = 1)
    //     (1 - tau_j) t_j(0) + tau_j t_j(1) = T_j     =>     t_j(1) = (T_j - (1 - tau_j) t_j(0)) / tau_j
    // -- a division by tau_j alone, known before the first round: ONE inversion per instance (Montgomery's trick over all taus), none
    // per round.  The same field elements as (claim - l(0) p t(0)) / (l(1) p) whenever p != 0 (claim = p T exactly); when a challenge
    // has zeroed p every s_j is the zero polynomial whatever t(1) is, so the reference's fall-back for that case (a third sum) is not
    // needed for the same output.  tau_j = 0 still takes the fall-back (t(1) is then not determined by the claim).
    std::vector<H> tau_inv;  // 1 / tau_i; zero where tau_i = 0
    mutable H T;             // T_round (valid from round 2 on: set by bound())
    mutable H lt0, lt1, ltinf;  // t(0), t(1), t(inf) of the round derive() last ran for
    mutable bool l1p_zero = false;  // the coming round takes the fall-back (tau_round = 0)
    void prepare() const {
      if (round > l) return;
      l1p_zero = taus[round - 1].is_zero();
    }
    void init(const ScAlg& a, const uint8_t* taus_bytes, uint32_t l_) {
      l = l_, first_half = l / 2, second_half = l - first_half, round = 1;
      taus.resize(l), eq0.resize(l), slope.resize(l), eqm1.resize(l), tau_inv.assign(l, H::zero());
      for (uint32_t i = 0; i < l; i++) {
        taus[i] = a.in(taus_bytes + 32 * (size_t)i);
        eq0[i] = H::one() - taus[i];
        slope[i] = taus[i] - eq0[i];
        eqm1[i] = eq0[i] - slope[i];
      }
      // all 1 / tau_i from one inversion: prefix products of the non-zero taus, invert the last, walk back
      std::vector<H> pre(l);
      H acc = H::one();
      for (uint32_t i = 0; i < l; i++) {
        pre[i] = acc;
        if (!taus[i].is_zero()) acc = acc * taus[i];
      }
      H inv = acc.inv();  // (acc is a product of non-zero elements, or ONE)
      for (uint32_t i = l; i-- > 0;) {
        if (taus[i].is_zero()) continue;
        tau_inv[i] = inv * pre[i];
        inv = inv * taus[i];
      }
      eval_eq_left = H::one();
    }
    // derive_from_claim_deg2 / _deg1 (sumcheck.rs:680-753): (s(0), cubic coefficient, s(-1)) from t(0), t(inf) and the claim;
    // t_m1() supplies t(-1) when tau = 0: the third N-scaling sum of the fallback_eval_inf_* paths (sumcheck.rs:1085-1222)
    template <class TM1> void derive(const H& t0, const H& tinf, const H& claim, bool deg1, H& s0, H& lead, H& sm1, TM1&& t_m1) const {
      const H& p = eval_eq_left;
      const H l0p = eq0[round - 1] * p;
      s0 = l0p * t0;
      lead = deg1 ? H::zero() : slope[round - 1] * p * tinf;
      H tm1, t1;
      prepare();
      if (!l1p_zero) {
        const H& Tj = round == 1 ? claim : T;  // (p_1 = 1: the caller's claim IS the inner claim)
        t1 = (Tj - eq0[round - 1] * t0) * tau_inv[round - 1];
        tm1 = t0.dbl() - t1;
        if (!deg1) tm1 = tm1 + tinf.dbl();  // t(-1) = 2 t(inf) + 2 t(0) - t(1)
      } else {
        tm1 = t_m1();
        t1 = t0.dbl() - tm1;
        if (!deg1) t1 = t1 + tinf.dbl();
      }
      lt0 = t0, lt1 = t1, ltinf = deg1 ? H::zero() : tinf;
      sm1 = eqm1[round - 1] * p * tm1;
    }
    void bound(const H& r) {  // sumcheck.rs:1226-1231
      const H& tau = taus[round - 1];
      eval_eq_left = eval_eq_left * (H::one() - tau - r + (r * tau).dbl());
      T = lt0 + r * ((lt1 - lt0 - ltinf) + r * ltinf);  // t_round(r): the next round's inner claim
      round++;
    }
    // eq over the variables still free in round `rnd` (1-based): taus[rnd .. l), most significant first -- what
    // poly_eq_left[first_half - rnd][id >> second_half] * poly_eq_right[second_half][id & mask] (first half) and
    // poly_eq_right[l - rnd][id] (last half) both spell out (sumcheck.rs:1233-1253)
    std::vector<H> factors(uint32_t rnd) const {
      std::vector<H> t(1, H::one());
      for (uint32_t i = l; i-- > rnd;) {  // eq.rs:54-73: for r in r.iter().rev()
        const size_t size = t.size();
        t.resize(2 * size);
        for (size_t x = 0; x < size; x++) {
          const H y = t[x] * taus[i];
          t[x + size] = y;
          t[x] = t[x] - y;
        }
      }
      return t;
    }
  };

  // ---- tail rounds over host tables ------------------------------------------------------------------------------------
  static void bind_top(std::vector<H>& z, const H& r) {  // multilinear.rs:65-84
    const size_t n = z.size() / 2;
    for (size_t i = 0; i < n; i++) z[i] = z[i] + r * (z[i + n] - z[i]);
    z.resize(n);
  }
  // (t(0), t(inf)) of evaluation_points_cubic_with_three_inputs (mode 3) / quadratic_with_one_input (mode 1); with m1: t(-1),
  // the fallback's third sum, in *t0
  static void eq_sums(int mode, bool m1, const std::vector<H>& A, const std::vector<H>& B, const std::vector<H>& C, const std::vector<H>& fac,
                      H* t0, H* tinf) {
    const size_t h = A.size() / 2;
    H s0 = H::zero(), s1 = H::zero();
    for (size_t id = 0; id < h; id++) {
      if (!m1) {
        if (mode == 1) {
          s0 = s0 + A[id] * fac[id];
          continue;
        }
        s0 = s0 + (A[id] * B[id] - C[id]) * fac[id];
        s1 = s1 + (A[id + h] - A[id]) * (B[id + h] - B[id]) * fac[id];
      } else {
        const H ma = A[id].dbl() - A[id + h];
        if (mode == 1) {
          s0 = s0 + ma * fac[id];
          continue;
        }
        const H mb = B[id].dbl() - B[id + h], mc = C[id].dbl() - C[id + h];
        s0 = s0 + (ma * mb - mc) * fac[id];
      }
    }
    *t0 = s0;
    if (tinf) *tinf = s1;
  }
  static void quad_sums(const std::vector<H>& A, const std::vector<H>& B, H* e0, H* quad) {  // sumcheck.rs:163-186
    const size_t h = A.size() / 2;
    H s0 = H::zero(), s1 = H::zero();
    for (size_t i = 0; i < h; i++) {
      s0 = s0 + A[i] * B[i];
      s1 = s1 + (A[i + h] - A[i]) * (B[i + h] - B[i]);
    }
    *e0 = s0, *quad = s1;
  }
};

template <int FID>
typename ScAlg<FID>::H ScAlg<FID>::ask(TranscriptFn cb, void* ctx, const H* coeffs, uint32_t n, uint8_t* polys_out, uint8_t* r_out,
                                       double* cb_ms) const {
  uint8_t buf[4 * 32], ch[32];
  for (uint32_t i = 0; i < n; i++) out(coeffs[i], buf + 32 * i);
  if (polys_out) memcpy(polys_out, buf, 32 * (size_t)n);
  (void)cb_ms;
  const int rc = cb(ctx, buf, n, ch);
  if (rc != 0) throw ScFail{1, "the transcript callback failed (" + std::to_string(rc) + ")"};
  if (r_out) memcpy(r_out, ch, 32);
  return in(ch);
}

// The rounds j0 .. l of prove_cubic_with_three_inputs (mode 3, sumcheck.rs:446-507) or prove_quad_prod (mode 4,
// sumcheck.rs:199-249) over HOST tables of the current length (A.size() == 2^(l - j0 + 1)).  On return the tables hold one
// element each: the final claims.
template <int FID, int MODE>
void sc_tail_rounds(const ScAlg<FID>& alg, typename ScAlg<FID>::Eq* eq, uint32_t l, uint32_t j0, typename ScAlg<FID>::H& claim,
                    std::vector<typename ScAlg<FID>::H>& A, std::vector<typename ScAlg<FID>::H>& B, std::vector<typename ScAlg<FID>::H>& C,
                    TranscriptFn cb, void* cb_ctx, uint8_t* out_polys, uint8_t* out_r) {
  using H = typename ScAlg<FID>::H;
  constexpr uint32_t NCO = MODE == 3 ? 4u : 3u;
  for (uint32_t j = j0; j <= l; j++) {
    H co[4];
    if (MODE == 3) {
      const std::vector<H> fac = eq->factors(j);
      H t0, tinf, s0, lead, sm1;
      ScAlg<FID>::eq_sums(3, false, A, B, C, fac, &t0, &tinf);
      eq->derive(t0, tinf, claim, false, s0, lead, sm1, [&] {
        H tm1;
        ScAlg<FID>::eq_sums(3, true, A, B, C, fac, &tm1, nullptr);
        return tm1;
      });
      ScAlg<FID>::from_evals_deg3(s0, claim, lead, sm1, co);
    } else {
      H e0, quad;
      ScAlg<FID>::quad_sums(A, B, &e0, &quad);
      ScAlg<FID>::from_evals_deg2(e0, claim, quad, co);
    }
    const H r = alg.ask(cb, cb_ctx, co, NCO, out_polys ? out_polys + 32 * NCO * (size_t)(j - 1) : nullptr, out_r ? out_r + 32 * (size_t)(j - 1) : nullptr);
    claim = ScAlg<FID>::poly_eval(co, NCO, r);
    ScAlg<FID>::bind_top(A, r), ScAlg<FID>::bind_top(B, r);
    if (MODE == 3) {
      ScAlg<FID>::bind_top(C, r);
      eq->bound(r);
    }
  }
}

// One claim of prove_batch_eval (sumcheck.rs:251-353) as the batch loop sees it: its round sums come from the device while the
// polynomial is long and from here once it is on the host.
template <int FID> struct ScBatchClaim {
  using H = typename ScAlg<FID>::H;
  typename ScAlg<FID>::Eq eq;
  H claim0, running, coeff;
  uint32_t num_rounds = 0;
  std::vector<H> host;  // the polynomial once it lives on the host (empty: still on the device)
  // evaluation points [e0, 0, em1] of this round from t(0) (sumcheck.rs:1039-1083)
  template <class TM1> void points(const H& t0, H& e0, H& em1, TM1&& t_m1) const {
    H lead;
    eq.derive(t0, H::zero(), running, true, e0, lead, em1, t_m1);
  }
  void host_points(H& e0, H& em1) const {
    const std::vector<H> fac = eq.factors(eq.round);
    const std::vector<H> none;
    H t0;
    ScAlg<FID>::eq_sums(1, false, host, none, none, fac, &t0, nullptr);
    points(t0, e0, em1, [&] {
      H tm1;
      ScAlg<FID>::eq_sums(1, true, host, none, none, fac, &tm1, nullptr);
      return tm1;
    });
  }
};

// prove_batch_eval's round loop (sumcheck.rs:281-353).  DEV supplies what the device does for claims whose polynomial is still
// in HBM -- the tail never calls it:
//   void start(size_t i)                       enqueue the first sums of claim i (unbound table, round 1 of its instance)
//   H    t0(size_t i)                          wait for claim i's pending sums, t(0) as an element
//   H    t_m1(size_t i)                        the fallback's third sum t(-1) over claim i's current table
//   void bind(size_t i, const H& r)            bind claim i's table with r and enqueue the next round's sums -- or, when the
//                                              bound table fits the tail, bring it to the host: then claims[i].host is filled
//   void ahead(size_t i)                       called once per round for every claim whose polynomial is still on the device, after its
//                                              eq instance has moved on to the coming round and prepared it (Eq::l1p_zero is valid): the
//                                              device may get the pass AFTER the pending one under way (sumcheck_prove.hpp prelaunch)
// out_finals (k x 32 bytes, the vectors' own form) = poly_finals.
template <int FID, class DEV>
void sc_batch_rounds(const ScAlg<FID>& alg, std::vector<ScBatchClaim<FID>>& claims, DEV& dev, TranscriptFn cb, void* cb_ctx,
                     uint8_t* out_polys, uint8_t* out_r, uint8_t* out_finals, uint32_t* rounds_done = nullptr) {
  using H = typename ScAlg<FID>::H;
  const size_t k = claims.size();
  uint32_t nmax = 0;
  for (const auto& c : claims) nmax = c.num_rounds > nmax ? c.num_rounds : nmax;
  H e = H::zero();  // (:281-289) e = sum claim_i 2^(nmax - n_i) coeff_i
  for (const auto& c : claims) e = e + c.claim0 * ScAlg<FID>::pow2(nmax - c.num_rounds) * c.coeff;
  for (size_t i = 0; i < k; i++)
    if (claims[i].num_rounds == nmax && claims[i].host.empty()) dev.start(i);
  for (size_t i = 0; i < k; i++)
    if (claims[i].num_rounds == nmax) claims[i].eq.prepare();  // round 1's inversions, under the passes just enqueued
  std::vector<H> e0(k), em1(k);
  for (uint32_t round = 0; round < nmax; round++) {
    const uint32_t remaining = nmax - round;
    for (size_t i = 0; i < k; i++) {
      ScBatchClaim<FID>& c = claims[i];
      if (remaining <= c.num_rounds) {  // (:301-305)
        if (!c.host.empty()) c.host_points(e0[i], em1[i]);
        else c.points(dev.t0(i), e0[i], em1[i], [&] { return dev.t_m1(i); });
      } else {  // not yet started: constant (:306-312)
        e0[i] = em1[i] = ScAlg<FID>::pow2(remaining - c.num_rounds - 1) * c.claim0;
      }
    }
    H c0 = H::zero(), cm1 = H::zero();
    for (size_t i = 0; i < k; i++) c0 = c0 + e0[i] * claims[i].coeff, cm1 = cm1 + em1[i] * claims[i].coeff;
    const H c1 = e - c0;
    const H quad = (c1 + cm1 - c0.dbl()) * ScAlg<FID>::two_inv();  // (S(1) + S(-1) - 2 S(0)) / 2
    H poly[3];
    ScAlg<FID>::from_evals_deg2(c0, e, quad, poly);
    const H r = alg.ask(cb, cb_ctx, poly, 3, out_polys ? out_polys + 96 * (size_t)round : nullptr, out_r ? out_r + 32 * (size_t)round : nullptr);
    if (rounds_done) ++*rounds_done;
    for (size_t i = 0; i < k; i++) {
      ScBatchClaim<FID>& c = claims[i];
      if (remaining <= c.num_rounds) {
        c.running = ScAlg<FID>::update_claim(c.running, e0[i], em1[i], r);
        if (!c.host.empty()) ScAlg<FID>::bind_top(c.host, r);
        else dev.bind(i, r);
        c.eq.bound(r);
      } else if (remaining - 1 == c.num_rounds && c.host.empty()) {  // joins in the next round
        dev.start(i);
      }
    }
    e = ScAlg<FID>::poly_eval(poly, 3, r);
    // the next round's inversions while the device runs the passes enqueued above
    // A pass may only go out ahead of its challenge when NO active claim takes the tau = 0 fall-back in the coming round: the
    // fall-back's extra pass allocates, copies and waits for a stream, and a device-wide wait behind a kernel that is itself
    // waiting for this thread's challenge never returns (the armed pass timed out and the proof failed with NMX_E_HIP).
    bool fallback_next = false;
    for (size_t i = 0; i < k; i++)
      if (remaining - 1 <= claims[i].num_rounds && remaining > 1) {
        claims[i].eq.prepare();
        fallback_next = fallback_next || claims[i].eq.l1p_zero;
      }
    for (size_t i = 0; i < k && !fallback_next; i++)
      if (remaining - 1 <= claims[i].num_rounds && remaining > 1 && claims[i].host.empty())
        dev.ahead(i);  // (the device may enqueue the pass after the pending one now: its challenge comes later)
  }
  if (out_finals)
    for (size_t i = 0; i < k; i++) alg.out(claims[i].host[0], out_finals + 32 * i);  // every polynomial ends on the host (len 1)
}

}  // namespace nmx
