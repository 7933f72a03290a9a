// sumcheck_prove.hpp -- Spartan's three sum-check provers as ONE C call each (round 5; BASELINE.json configs[4], the sum-check
// half of `RelaxedR1CSSNARK::prove`, /root/reference/src/spartan/snark.rs:113-260).  Included at the end of sumcheck.hip.
//
//   prove_cubic_with_three_inputs   src/spartan/sumcheck.rs:446-507   (outer: eq(tau, x) (Az Bz - uCz_E)(x))
//   prove_quad_prod                 src/spartan/sumcheck.rs:199-249   (inner: ABC(y) z(y))
//   prove_batch_eval                src/spartan/sumcheck.rs:251-353   (batch_eval_reduce, src/spartan/mod.rs:377-437)
//   EqSumCheckInstance              src/spartan/sumcheck.rs:593-1253  (eq tables, claim-derived evaluation points, bound)
//
// Why one call per prover and not one call per round: a round is a streaming pass whose size halves every time, followed by a
// challenge that only the host's transcript can produce.  At n = 2^20 the passes of all 20 rounds move ~0.3 GB (~0.1 ms); what is
// left is 20 x (launch + result + host algebra + transcript).  The per-round C calls (nmx_sumcheck_eq_sums / _bind_eq_sums) pay a
// context lease, two launches, a device-to-host copy and a stream synchronisation each -- and the caller's FFI crossing.  Here
// the round loop lives behind the boundary:
//   * the tables are bound IN PLACE and the bind of round j is fused with the sums of round j + 1 (k_bind_eq_sums and its
//     two-vector sibling below): every table is read once per round;
//   * the sums land in a MAILBOX -- a few words of coherent pinned host memory the last block writes with system scope, sequence
//     word last -- which the host polls: no copy engine, no stream synchronisation in a round;
//   * rounds whose bound half fits one block (<= 512 indices) run as ONE launch (k_sc_small) instead of pass + final sum;
//   * the O(1) algebra of a round (derive_from_claim_deg2/1, UniPoly::from_evals_deg3/2, evaluate, EqSumCheckInstance::bound)
//     runs on the host in the library's own field arithmetic; the only thing that leaves the library is the transcript step:
//     a callback receives the round polynomial's coefficients and returns the challenge
//     (`transcript.absorb(b"p", &poly); transcript.squeeze(b"c")`, sumcheck.rs:224-227,481-484,315-318 -- Keccak stays in Rust).
// All sqrt-size eq tables of an instance (poly_eq_left[k], poly_eq_right[k], sumcheck.rs:608-641) are built by one launch into a
// heap layout (table k at offset 2^k) in the context's aux arena.
#pragma once

namespace nmx {

// ---- mailbox ----------------------------------------------------------------------------------------------------------
static constexpr uint32_t kMailSlots = 16, kMailSlotWords = 64;  // slot = 256 bytes: word 0 sequence, words 8..31 three field elements
static constexpr uint32_t kScSmallHq = 512;                       // bound halves up to this many indices run as one block

__device__ __forceinline__ void mail_publish(uint32_t* slot, uint32_t seq) {
  __threadfence_system();  // the result words (plain stores into host memory) before the sequence word
  __hip_atomic_store(slot, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// final sum of per-block partials straight into a mailbox slot (k_sum_partials_n with a different destination)
template <int FID, int J, int STRIDE>
__global__ __launch_bounds__(256) void k_sum_partials_mail(const uint32_t* partial, uint32_t nparts, uint32_t* slot, uint32_t seq) {
  using F = Fp<FID>;
  __shared__ uint32_t lds[36 * J];
  F s[J];
#pragma unroll
  for (int j = 0; j < J; j++) s[j] = F::zero();
  uint32_t pending = 0;
  for (uint32_t i = threadIdx.x; i < nparts; i += 256) {
#pragma unroll
    for (int j = 0; j < J; j++) s[j] = (s[j] + ldw<FID>(partial, STRIDE * (size_t)i + j)).norm();
    if (++pending == 8) {
#pragma unroll
      for (int j = 0; j < J; j++) s[j] = s[j].canon();
      pending = 0;
    }
  }
#pragma unroll
  for (int j = 0; j < J; j++) s[j] = s[j].canon();
  block_sum_waves<FID, J>(s, lds);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int j = 0; j < J; j++) s[j].to_words(slot + 8 + 8 * j);
    mail_publish(slot, seq);
  }
}

// ---- quad_prod: bind two tables with the round challenge AND the next round's sums (no eq factor) -----------------------
// compute_eval_points_quad_prod (sumcheck.rs:163-186) over the tables bound by bind_poly_var_top (multilinear.rs:65-84) in
// the same pass: 256 B per index (four reads, two writes per table pair) instead of 192 B (binds) + 128 B (sums).
template <int FID> struct BindQpArgs {
  const uint32_t *A, *B;
  uint32_t *oA, *oB;
  Fp<FID> r;
  uint32_t hq;
};
template <int FID>
__device__ __forceinline__ void sc_bind2(const uint32_t* X, uint32_t* oX, const Fp<FID>& r, uint32_t id, uint32_t hq, Fp<FID>& y0,
                                         Fp<FID>& y1) {
  using F = Fp<FID>;
  const F x00 = ldw<FID>(X, id), x01 = ldw<FID>(X, (size_t)id + hq);
  const F x10 = ldw<FID>(X, (size_t)id + 2 * (size_t)hq), x11 = ldw<FID>(X, (size_t)id + 3 * (size_t)hq);
  y0 = (x00 + r * F::sub2(x10, x00).norm()).norm().canon();  // lo + r (hi - lo), as BindTopFn
  y1 = (x01 + r * F::sub2(x11, x01).norm()).norm().canon();
  y0.to_words(oX + 8 * (size_t)id);
  y1.to_words(oX + 8 * ((size_t)id + hq));
}
template <int FID> __global__ __launch_bounds__(256) void k_bind_qp_sums(BindQpArgs<FID> a, uint32_t* partial) {
  using F = Fp<FID>;
  __shared__ uint32_t lds[72];
  F s0 = F::zero(), s1 = F::zero();
  uint32_t pending = 0;
  for (uint32_t id = blockIdx.x * 256u + threadIdx.x; id < a.hq; id += gridDim.x * 256u) {
    F a0, a1, b0, b1;
    sc_bind2<FID>(a.A, a.oA, a.r, id, a.hq, a0, a1);
    sc_bind2<FID>(a.B, a.oB, a.r, id, a.hq, b0, b1);
    s0 = s0 + a0 * b0;
    s1 = s1 + F::sub2(a1, a0).norm() * F::sub2(b1, b0).norm();
    if (++pending == 6) {
      s0 = s0.norm().canon();
      s1 = s1.norm().canon();
      pending = 0;
    }
  }
  s0 = s0.norm().canon();
  s1 = s1.norm().canon();
  block_sum_pair<FID>(s0, s1, lds);
  if (threadIdx.x == 0) {
    s0.to_words(partial + 16 * blockIdx.x);
    s1.to_words(partial + 16 * blockIdx.x + 8);
  }
}

// ---- one block: bind + next sums + mailbox, for bound halves of <= kScSmallHq indices ------------------------------------
// MODE 1 / 3: the eq-factored rounds (as k_bind_eq_sums); MODE 4: quad_prod.  bind = 0: sums only (a first round that is small).
template <int FID> struct ScSmallArgs {
  const uint32_t *A, *B, *C;
  uint32_t *oA, *oB, *oC;
  const uint32_t *eqL, *eqR;
  Fp<FID> r, nk;
  uint32_t shift, mask, hq, bind, seq;
  uint32_t* slot;
};
template <int FID, int MODE> __global__ __launch_bounds__(256) void k_sc_small(ScSmallArgs<FID> a) {
  using F = Fp<FID>;
  __shared__ uint32_t lds[72];
  F s0 = F::zero(), s1 = F::zero();
  uint32_t pending = 0;
  for (uint32_t id = threadIdx.x; id < a.hq; id += 256u) {
    F a0, a1, b0 = F::zero(), b1 = F::zero(), c0 = F::zero(), c1;
    if (a.bind) {
      sc_bind2<FID>(a.A, a.oA, a.r, id, a.hq, a0, a1);
      if (MODE >= 3) sc_bind2<FID>(a.B, a.oB, a.r, id, a.hq, b0, b1);
      if (MODE == 3) sc_bind2<FID>(a.C, a.oC, a.r, id, a.hq, c0, c1);
    } else {
      a0 = ldw<FID>(a.A, id), a1 = ldw<FID>(a.A, (size_t)id + a.hq);
      if (MODE >= 3) b0 = ldw<FID>(a.B, id), b1 = ldw<FID>(a.B, (size_t)id + a.hq);
      if (MODE == 3) c0 = ldw<FID>(a.C, id);
    }
    if (MODE == 4) {
      s0 = s0 + a0 * b0;
      s1 = s1 + F::sub2(a1, a0).norm() * F::sub2(b1, b0).norm();
    } else {
      F fac = ldw<FID>(a.eqR, a.eqL ? (id & a.mask) : id);
      if (a.eqL) fac = ldw<FID>(a.eqL, id >> a.shift) * fac;
      if (MODE == 1) {
        s0 = s0 + a0 * fac;
      } else {
        const F e0 = F::mul_add(a0, b0, c0, a.nk);  // a0 b0 - c0 k in one reduction (nk = p - k)
        const F q = F::sub2(a1, a0).norm() * F::sub2(b1, b0).norm();
        s0 = s0 + e0 * fac;
        s1 = s1 + q * fac;
      }
    }
    if (++pending == 6) {
      s0 = s0.norm().canon();
      s1 = s1.norm().canon();
      pending = 0;
    }
  }
  s0 = s0.norm().canon();
  s1 = s1.norm().canon();
  block_sum_pair<FID>(s0, s1, lds);
  if (threadIdx.x == 0) {
    s0.to_words(a.slot + 8);
    s1.to_words(a.slot + 16);
    mail_publish(a.slot, a.seq);
  }
}

// ---- the last bind: tables of two elements -> one; the values ARE the final claims (poly_A[0], ..., sumcheck.rs:241-248) ---
struct ScFinalArgs {
  uint32_t* X[3];
  uint32_t n, seq;
  uint32_t* slot;
};
template <int FID> __global__ __launch_bounds__(64) void k_sc_final(ScFinalArgs a, Fp<FID> r) {
  using F = Fp<FID>;
  const uint32_t t = threadIdx.x;
  if (t < a.n) {
    const F x0 = ldw<FID>(a.X[t], 0), x1 = ldw<FID>(a.X[t], 1);
    const F y = (x0 + r * F::sub2(x1, x0).norm()).norm().canon();
    y.to_words(a.X[t]);
    y.to_words(a.slot + 8 + 8 * t);
    __threadfence_system();
  }
  __syncthreads();
  if (t == 0) mail_publish(a.slot, a.seq);
}

// ---- all eq tables of one instance in one launch ---------------------------------------------------------------------
// poly_eq_left[k] / poly_eq_right[k] (sumcheck.rs:612-641) = eq over the LAST k challenges of their side, most significant
// variable first.  Heap layout per side: entry g in [1, 2^(K+1)) holds table k = floor(log2 g) at x = g - 2^k.
template <int FID> struct EqHeapFn {
  static constexpr uint32_t kMaxEll = 12;
  uint32_t *heapL, *heapR;
  Fp<FID> r[2 * kMaxEll], nr[2 * kMaxEll];  // tau_i 2^261 and (1 - tau_i) 2^261: the left side's challenges, then the right side's
  Fp<FID> one;                              // ONE in the vectors' form
  uint32_t KL, KR;
  NMX_HD void operator()(uint32_t g) const {
    const uint32_t nl = 2u << KL;
    const bool right = g >= nl;
    const uint32_t gg = right ? g - nl : g;
    if (gg == 0) return;
    uint32_t k = 0;
    while ((2u << k) <= gg) k++;
    const uint32_t x = gg - (1u << k), K = right ? KR : KL, o = (right ? KL : 0u) + (K - k);
    Fp<FID> acc = one;
    for (uint32_t i = 0; i < k; i++) acc = acc * (((x >> (k - 1 - i)) & 1u) ? r[o + i] : nr[o + i]);
    st_words(right ? heapR : heapL, gg, acc);
  }
  static NMX_HD void st_words(uint32_t* p, size_t i, const Fp<FID>& v) { v.canon().to_words(p + 8 * i); }
};
// eq.rs:54-73, one doubling step in place (fieldvec.hip's EqStepFn lives in another translation unit)
template <int FID> struct ScEqStepFn {
  uint32_t* buf;
  Fp<FID> r;
  uint32_t size;
  NMX_HD void operator()(uint32_t i) const {
    using F = Fp<FID>;
    const F x = F::from_words(buf + 8 * (size_t)i);
    const F y = r * x;
    y.canon().to_words(buf + 8 * ((size_t)i + size));
    F::sub2(x, y).norm().canon().to_words(buf + 8 * (size_t)i);
  }
};
// table k from table k + 1 (sides with more than kMaxEll challenges): T_k[x] = T_{k+1}[x] + T_{k+1}[x + 2^k]
template <int FID> struct EqHalveFn {
  uint32_t* heap;
  uint32_t k;
  NMX_HD void operator()(uint32_t x) const {
    using F = Fp<FID>;
    const size_t src = (size_t)2 << k;
    const F v = (F::from_words(heap + 8 * (src + x)) + F::from_words(heap + 8 * (src + x + ((size_t)1 << k)))).norm().canon();
    v.to_words(heap + 8 * (((size_t)1 << k) + x));
  }
};

// ---- host side ----------------------------------------------------------------------------------------------------------
// the transcript's side of a round: TranscriptFn (runtime.hpp; include/nova_mi355x.h nmx_transcript_fn)
struct ScProf {  // wall-clock split of one prover call (profiling on): where a round's time goes
  double wait = 0, host = 0, cb = 0;
  uint32_t launches = 0, rounds = 0;
};

template <int FID> struct ScHost {
  using F = Fp<FID>;
  Ctx& c;
  bool mont;
  F corr[5];   // raw device sums -> the vectors' own form, by the number of stored factors per term (eq_sums_t's constants)
  F to_mont;   // 2^256 as a plain residue: internal -> Montgomery words
  ScProf prof;
  bool profiling;
  ScHost(Ctx& ctx, uint32_t flags) : c(ctx), mont((flags & NMX_SCALARS_MONT) != 0), profiling(G.profiling) {
    for (uint32_t k = 1; k <= 4; k++) corr[k] = pow2_plain<FID>(261u * k - (mont ? 256u * (k - 1) : 0u));
    to_mont = pow2_plain<FID>(256);
    mail_init();
  }
  // --- canonical host arithmetic on internal residues (every value < p after every operation)
  static F add(const F& a, const F& b) { return (a + b).norm().canon(); }
  static F sub(const F& a, const F& b) { return F::sub2(a, b).norm().canon(); }
  static F mul(const F& a, const F& b) { return (a * b).canon(); }
  static F dbl(const F& a) { return add(a, a); }
  static bool is_zero(const F& a) { return a.canon().is_zero_limbs(); }
  F in(const void* p) const { return challenge_internal<FID>(p, mont); }
  void out(const F& v, uint8_t* p) const {
    uint32_t w[8];
    (mont ? (v * to_mont).canon() : v.to_canonical()).to_words(w);
    memcpy(p, w, 32);
  }
  // a raw device sum of terms with k stored factors -> internal residue
  F raw(const uint32_t* words, uint32_t k) const {
    const F v = (F::from_words(words) * corr[k]).canon();  // x * Fm: the vectors' own form
    return (mont ? v.mont256_to_internal() : v.to_internal()).canon();
  }
  // a stored element (the vectors' own form) read back from the device -> internal
  F stored(const uint32_t* words) const {
    const F v = F::from_words(words);
    return (mont ? v.mont256_to_internal() : v.to_internal()).canon();
  }

  // --- mailbox
  void mail_init() {
    if (c.mail) return;
    void* p = nullptr;
    HIPCHK(hipHostMalloc(&p, kMailSlots * kMailSlotWords * 4, hipHostMallocCoherent | hipHostMallocMapped));
    memset(p, 0, kMailSlots * kMailSlotWords * 4);
    void* d = nullptr;
    HIPCHK(hipHostGetDevicePointer(&d, p, 0));
    c.mail = (char*)p;
    c.mail_dev = (char*)d;
  }
  uint32_t* slot_dev(uint32_t s) const { return (uint32_t*)c.mail_dev + (size_t)s * kMailSlotWords; }
  uint32_t next_seq() { return ++c.mail_seq ? c.mail_seq : ++c.mail_seq; }  // never 0 (a fresh mailbox reads 0)
  // waits until slot s carries `seq`; returns its three result words blocks.  Polling the sequence word costs a PCIe read
  // of host memory by the host itself (none); the stream is only synchronised when the poll gives up or polling is off.
  const uint32_t* wait(uint32_t s, uint32_t seq) {
    const auto t0 = std::chrono::steady_clock::now();
    uint32_t* host = (uint32_t*)c.mail + (size_t)s * kMailSlotWords;
    const uint32_t poll_us = G.sc_poll_us.load(std::memory_order_relaxed);
    bool ok = false;
    if (poll_us) {
      for (uint32_t spin = 0;; spin++) {
        if (__atomic_load_n(host, __ATOMIC_ACQUIRE) == seq) {
          ok = true;
          break;
        }
        if ((spin & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(poll_us)) break;
      }
    }
    if (!ok) {
      stream_wait(c.stream);
      require(__atomic_load_n(host, __ATOMIC_ACQUIRE) == seq, NMX_E_HIP, "sum-check: the round's result never reached its mailbox");
    }
    if (profiling) prof.wait += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return host + 8;
  }
  void launched(uint32_t n = 1) { prof.launches += n; }
  // transcript step: coefficients out (the vectors' own form), challenge in
  F ask(TranscriptFn cb, void* ctx, const F* coeffs, uint32_t n, uint8_t* polys_out, uint8_t* r_out) {
    uint8_t buf[4 * 32], ch[32];
    for (uint32_t i = 0; i < n; i++) out(coeffs[i], buf + 32 * i);
    if (polys_out) memcpy(polys_out, buf, 32 * (size_t)n);
    const auto t0 = std::chrono::steady_clock::now();
    const int rc = cb(ctx, buf, n, ch);
    if (profiling) prof.cb += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (rc != 0) throw Fail{NMX_E_ARG, "sum-check: the transcript callback failed (" + std::to_string(rc) + ")"};
    if (r_out) memcpy(r_out, ch, 32);
    return in(ch);  // NMX_E_SCALAR_RANGE for a challenge >= p
  }
  // UniPoly::evaluate (univariate.rs:140-149)
  static F poly_eval(const F* co, uint32_t n, const F& r) {
    F eval = co[0], power = r;
    for (uint32_t i = 1; i < n; i++) {
      eval = add(eval, mul(power, co[i]));
      power = mul(power, r);
    }
    return eval;
  }
  F two_inv() const {
    F two = F::zero();
    two.l[0] = 2;
    return two.to_internal().canon().inv().canon();
  }
  F pow2(uint32_t e) const {  // Scalar::from(2).pow_vartime([e]), internal
    F two = F::zero();
    two.l[0] = 2;
    const F t = two.to_internal().canon();
    F acc = F::one();
    for (uint32_t i = 0; i < e; i++) acc = mul(acc, t);
    return acc;
  }
  void finish_profile(double total_ms) {
    if (!profiling) return;
    float v[6] = {(float)total_ms, (float)prof.wait, (float)(total_ms - prof.wait - prof.cb), (float)prof.cb, (float)prof.launches,
                  (float)prof.rounds};
    prof_store(v, 6);
  }
};

// grids of the passes (the rules of eq_sums_t / bind_eq_sums_t / plain_sums_t)
static inline uint32_t sc_blocks_sums(uint32_t h, bool mode1) {
  const uint32_t want = (h + 256 * 8 - 1) / (256 * 8), cap = G.eq_max_blocks ? (uint32_t)G.eq_max_blocks : (mode1 ? 768u : 2048u);
  return want < 1 ? 1 : (want > cap ? cap : want);
}
static inline uint32_t sc_blocks_bind(uint32_t hq) {
  const uint32_t want = (hq + 256 * 4 - 1) / (256 * 4);
  return want < 1 ? 1 : (want > 4096 ? 4096 : want);
}
static constexpr size_t kScPartialBytes = 4096 * 128;  // per mailbox slot: 4096 blocks x up to 32 words

// EqSumCheckInstance (sumcheck.rs:593-677, 1226-1253): host scalars + the device heaps of eq tables
template <int FID> struct ScEq {
  using F = Fp<FID>;
  using H = ScHost<FID>;
  uint32_t l = 0, first_half = 0, second_half = 0, round = 1;
  std::vector<F> taus, eq0, slope, eqm1;
  F eval_eq_left;
  uint32_t *heapL = nullptr, *heapR = nullptr;  // device
  uint32_t KL = 0, KR = 0;
  static size_t heap_bytes(uint32_t l_) {
    const uint32_t fh = l_ / 2, sh = l_ - fh, kl = fh > 0 ? fh - 1 : 0;
    return (((size_t)2 << kl) + ((size_t)2 << sh)) * 32 + 512;
  }
  // builds the tables on c.stream (no wait) into [mem, mem + heap_bytes(l))
  void init(H& h, const uint8_t* taus_bytes, uint32_t l_, char* mem, uint32_t flags) {
    l = l_, first_half = l / 2, second_half = l - first_half, round = 1;
    KL = first_half > 0 ? first_half - 1 : 0, KR = second_half;
    taus.resize(l), eq0.resize(l), slope.resize(l), eqm1.resize(l);
    const F one = F::one();
    for (uint32_t i = 0; i < l; i++) {
      taus[i] = h.in(taus_bytes + 32 * (size_t)i);
      eq0[i] = H::sub(one, taus[i]);              // eq(tau, 0)
      slope[i] = H::sub(taus[i], eq0[i]);         // 2 tau - 1
      eqm1[i] = H::sub(eq0[i], slope[i]);         // eq(tau, -1) = 2 - 3 tau
    }
    eval_eq_left = one;
    heapL = (uint32_t*)mem;
    heapR = (uint32_t*)(mem + ((((size_t)2 << KL) * 32 + 255) & ~(size_t)255));
    DeviceBackend be(h.c, false, false);
    // ONE in the vectors' form (as eq_evals_t)
    uint32_t w[8];
    F onev = F::zero();
    onev.l[0] = 1;
    if (h.mont) h.to_mont.to_words(w);
    else onev.to_words(w);
    if (KL <= EqHeapFn<FID>::kMaxEll && KR <= EqHeapFn<FID>::kMaxEll) {
      EqHeapFn<FID> f;
      f.heapL = heapL, f.heapR = heapR, f.KL = KL, f.KR = KR, f.one = F::from_words(w);
      for (uint32_t i = 0; i < 2 * EqHeapFn<FID>::kMaxEll; i++) f.r[i] = f.nr[i] = F::zero();
      // left side: taus[1 .. first_half) (sumcheck.rs:634-635: skip(1)); right side: taus[first_half .. l)
      for (uint32_t i = 0; i < KL; i++) f.r[i] = taus[1 + i], f.nr[i] = eq0[1 + i];
      for (uint32_t i = 0; i < KR; i++) f.r[KL + i] = taus[first_half + i], f.nr[KL + i] = eq0[first_half + i];
      be.launch(f, (2u << KL) + (2u << KR));
      h.launched();
    } else {  // long sides: the largest table by doubling (eq.rs:54-73), the others by pairwise sums
      auto side = [&](uint32_t* heap, uint32_t K, uint32_t first_tau) {
        HIPCHK(hipMemcpyAsync(heap + 8, w, 32, hipMemcpyHostToDevice, h.c.stream));  // table 0 = [ONE]
        stream_wait(h.c.stream);                                                     // w is a stack buffer
        if (K == 0) return;
        uint32_t* top = heap + 8 * ((size_t)1 << K);
        HIPCHK(hipMemcpyAsync(top, heap + 8, 32, hipMemcpyDeviceToDevice, h.c.stream));
        uint32_t size = 1;
        for (int j = (int)K - 1; j >= 0; j--) {
          ScEqStepFn<FID> f{top, taus[first_tau + (uint32_t)j], size};
          be.launch(f, size);
          size *= 2;
        }
        for (int k = (int)K - 1; k >= 1; k--) {
          EqHalveFn<FID> f{heap, (uint32_t)k};
          be.launch(f, 1u << k);
        }
        h.launched(2 * K);
      };
      side(heapL, KL, 1);
      side(heapR, KR, first_half);
    }
  }
  struct Tables {
    const uint32_t *eqL, *eqR;
    uint32_t shift, mask;
  };
  Tables tables(uint32_t rnd) const {  // poly_eqs_first_half / poly_eq_right_last_half (sumcheck.rs:1233-1253)
    if (rnd < first_half) {
      return Tables{heapL + 8 * ((size_t)1 << (first_half - rnd)), heapR + 8 * ((size_t)1 << second_half), second_half,
                    second_half >= 32 ? 0xffffffffu : ((1u << second_half) - 1u)};
    }
    return Tables{nullptr, heapR + 8 * ((size_t)1 << (l - rnd)), 0, 0xffffffffu};
  }
  // derive_from_claim_deg2 / _deg1 (sumcheck.rs:680-753); third() computes t(1) on the device when l(1) p = 0 (tau = 0 or a
  // challenge that zeroed eval_eq_left: the fallback_eval_inf_* paths, sumcheck.rs:1085-1222, whose third N-scaling sum is
  // t(-1) = 2 t(inf) + 2 t(0) - t(1) -- the same value from a sum over the HIGH halves).
  template <class Third> void derive(const F& t0, const F& tinf, const F& claim, bool deg1, F& s0, F& lead, F& sm1, Third&& third) const {
    const F& p = eval_eq_left;
    const F l0p = H::mul(eq0[round - 1], p), l1p = H::mul(H::add(eq0[round - 1], slope[round - 1]), p);
    s0 = H::mul(l0p, t0);
    lead = deg1 ? F::zero() : H::mul(H::mul(slope[round - 1], p), tinf);
    F t1;
    if (!H::is_zero(l1p)) t1 = H::mul(H::sub(claim, s0), l1p.inv().canon());
    else t1 = third();
    F tm1 = H::sub(H::dbl(t0), t1);
    if (!deg1) tm1 = H::add(tm1, H::dbl(tinf));
    sm1 = H::mul(H::mul(eqm1[round - 1], p), tm1);
  }
  void bound(const F& r) {  // sumcheck.rs:1226-1231
    const F& tau = taus[round - 1];
    F t = H::sub(H::sub(F::one(), tau), r);
    t = H::add(t, H::dbl(H::mul(r, tau)));
    eval_eq_left = H::mul(eval_eq_left, t);
    round++;
  }
};

// the launches of one eq-factored instance (MODE 1 or 3) or of quad_prod (MODE 4) over in-place tables
template <int FID, int MODE> struct ScPass {
  using F = Fp<FID>;
  using H = ScHost<FID>;
  H& h;
  uint32_t *A, *B, *C;
  uint32_t* partial;  // device scratch of this instance (kScPartialBytes)
  uint32_t slot;
  F nk;
  ScPass(H& h_, void* a, void* b, void* cc, uint32_t* partial_, uint32_t slot_)
      : h(h_), A((uint32_t*)a), B((uint32_t*)b), C((uint32_t*)cc), partial(partial_), slot(slot_) {
    F fconst = F::zero();
    if (MODE == 3) {
      if (h.mont) fconst = pow2_plain<FID>(256);
      else fconst.l[0] = 1;
    }
    nk = MODE == 3 ? F::sub2(F::zero(), fconst.canon()).norm().canon() : F::zero();  // p - k, as eq_sums_t
  }
  static constexpr uint32_t kFactors = MODE == 1 ? 2u : MODE == 3 ? 3u : 2u;  // stored factors per term without eqL
  uint32_t factors(const typename ScEq<FID>::Tables& t) const { return kFactors + (MODE != 4 && t.eqL ? 1u : 0u); }
  // sums only over tables of `len` elements (round 1, and the high-half sum of the fallback)
  uint32_t sums(const uint32_t* a, const uint32_t* b, const uint32_t* cc, size_t len, const typename ScEq<FID>::Tables& t) {
    const uint32_t hh = (uint32_t)(len / 2), seq = h.next_seq();
    hipStream_t s = h.c.stream;
    if (hh <= kScSmallHq) {
      ScSmallArgs<FID> x{a, b, cc, nullptr, nullptr, nullptr, t.eqL, t.eqR, F::zero(), nk, t.shift, t.mask, hh, 0u, seq, h.slot_dev(slot)};
      hipLaunchKernelGGL((k_sc_small<FID, MODE>), dim3(1), dim3(256), 0, s, x);
      h.launched();
    } else {
      const uint32_t blocks = sc_blocks_sums(hh, MODE == 1);
      if (MODE == 4) {
        hipLaunchKernelGGL((k_plain_sums<FID, 1>), dim3(blocks), dim3(256), 0, s, a, b, (const uint32_t*)nullptr, hh, partial);
        hipLaunchKernelGGL((k_sum_partials_mail<FID, 2, 4>), dim3(1), dim3(256), 0, s, partial, blocks, h.slot_dev(slot), seq);
      } else {
        constexpr int M = MODE == 4 ? 1 : MODE;
        if (t.eqL && t.shift < 31) hipLaunchKernelGGL((k_eq_rows<FID, M>), dim3(blocks), dim3(256), 0, s, a, b, cc, t.eqL, t.eqR, t.shift, hh, nk, partial);
        else hipLaunchKernelGGL((k_eq_sums<FID, M>), dim3(blocks), dim3(256), 0, s, a, b, cc, t.eqL, t.eqR, t.shift, t.mask, hh, nk, partial);
        hipLaunchKernelGGL((k_sum_partials_mail<FID, 2, 2>), dim3(1), dim3(256), 0, s, partial, blocks, h.slot_dev(slot), seq);
      }
      h.launched(2);
    }
    HIPCHK(hipGetLastError());
    return seq;
  }
  // bind the tables (len elements each) with r in place AND the sums of the next round over the bound halves
  uint32_t bind_sums(size_t len, const F& r, const typename ScEq<FID>::Tables& t) {
    const uint32_t hq = (uint32_t)(len / 4), seq = h.next_seq();
    hipStream_t s = h.c.stream;
    if (hq <= kScSmallHq) {
      ScSmallArgs<FID> x{A, B, C, A, B, C, t.eqL, t.eqR, r, nk, t.shift, t.mask, hq, 1u, seq, h.slot_dev(slot)};
      hipLaunchKernelGGL((k_sc_small<FID, MODE>), dim3(1), dim3(256), 0, s, x);
      h.launched();
    } else {
      const uint32_t blocks = sc_blocks_bind(hq);
      if (MODE == 4) {
        BindQpArgs<FID> x{A, B, A, B, r, hq};
        hipLaunchKernelGGL((k_bind_qp_sums<FID>), dim3(blocks), dim3(256), 0, s, x, partial);
      } else {
        constexpr int M = MODE == 4 ? 1 : MODE;
        hipLaunchKernelGGL((k_bind_eq_sums<FID, M>), dim3(blocks), dim3(256), 0, s, (const uint32_t*)A, (const uint32_t*)B,
                           (const uint32_t*)C, A, B, C, r, t.eqL, t.eqR, t.shift, t.mask, hq, nk, partial);
      }
      hipLaunchKernelGGL((k_sum_partials_mail<FID, 2, 2>), dim3(1), dim3(256), 0, s, partial, blocks, h.slot_dev(slot), seq);
      h.launched(2);
    }
    HIPCHK(hipGetLastError());
    return seq;
  }
  // t(1) of the current round: the sums pass over a copy of the tables with the halves swapped (the fallback; never on
  // transcript-derived challenges).  Returns the raw first sum.
  F high_half_sum(size_t len, const typename ScEq<FID>::Tables& t) {
    const size_t hb = len / 2 * 32;
    const int nt = MODE == 1 ? 1 : MODE == 3 ? 3 : 2;
    char* tmp = nullptr;
    HIPCHK(hipMalloc((void**)&tmp, (size_t)nt * len * 32));
    const uint32_t* src[3] = {A, B, C};
    try {
      for (int i = 0; i < nt; i++) {
        HIPCHK(hipMemcpyAsync(tmp + (size_t)i * len * 32, (const char*)src[i] + hb, hb, hipMemcpyDeviceToDevice, h.c.stream));
        HIPCHK(hipMemcpyAsync(tmp + (size_t)i * len * 32 + hb, src[i], hb, hipMemcpyDeviceToDevice, h.c.stream));
      }
      const uint32_t* ta = (const uint32_t*)tmp;
      const uint32_t seq = sums(ta, nt > 1 ? ta + 8 * len : nullptr, nt > 2 ? ta + 16 * len : nullptr, len, t);
      const F v = h.raw(h.wait(slot, seq), factors(t));
      stream_wait(h.c.stream);
      (void)hipFree(tmp);
      return v;
    } catch (...) {
      (void)hipStreamSynchronize(h.c.stream);
      (void)hipFree(tmp);
      throw;
    }
  }
};

template <int FID> static uint32_t sc_final(ScHost<FID>& h, uint32_t* const* X, uint32_t n, const Fp<FID>& r, uint32_t slot) {
  ScFinalArgs a;
  for (uint32_t i = 0; i < 3; i++) a.X[i] = i < n ? X[i] : nullptr;
  a.n = n, a.seq = h.next_seq(), a.slot = h.slot_dev(slot);
  hipLaunchKernelGGL((k_sc_final<FID>), dim3(1), dim3(64), 0, h.c.stream, a, r);
  HIPCHK(hipGetLastError());
  h.launched();
  return a.seq;
}

// SumcheckProof::prove_cubic_with_three_inputs (sumcheck.rs:446-507) / prove_quad_prod (sumcheck.rs:199-249)
template <int FID, int MODE>
static void sc_prove_t(Ctx& c, const void* claim, const void* taus, size_t num_rounds, void* A, void* B, void* C, uint32_t flags,
                       TranscriptFn cb, void* cb_ctx, uint8_t* out_polys, uint8_t* out_r, uint8_t* out_claims) {
  using F = Fp<FID>;
  using H = ScHost<FID>;
  const auto T0 = std::chrono::steady_clock::now();
  constexpr uint32_t NCO = MODE == 3 ? 4u : 3u, NT = MODE == 3 ? 3u : 2u;
  const uint32_t l = (uint32_t)num_rounds;
  H h(c, flags);
  arena_reserve(c, kScPartialBytes + 512);
  ScEq<FID> eq;
  if (MODE == 3) {
    aux_reserve(c, ScEq<FID>::heap_bytes(l));
    eq.init(h, (const uint8_t*)taus, l, c.aux, flags);
  }
  ScPass<FID, MODE> pass(h, A, B, C, (uint32_t*)c.arena, 0);
  F claim_i = h.in(claim);
  const F tinv = h.two_inv();
  size_t len = (size_t)1 << l;
  typename ScEq<FID>::Tables tb = MODE == 3 ? eq.tables(1) : typename ScEq<FID>::Tables{nullptr, nullptr, 0, 0};
  uint32_t seq = l ? pass.sums(pass.A, pass.B, pass.C, len, tb) : 0u;
  for (uint32_t j = 1; j <= l; j++) {
    const uint32_t* res = h.wait(0, seq);
    const F t0 = h.raw(res, pass.factors(tb)), t1 = h.raw(res + 8, pass.factors(tb));
    F co[4];
    if (MODE == 3) {
      F s0, lead, sm1;
      eq.derive(t0, t1, claim_i, false, s0, lead, sm1, [&] { return pass.high_half_sum(len, tb); });
      const F s1 = H::sub(claim_i, s0);  // evals = [s(0), claim - s(0), cubic coefficient, s(-1)]
      co[0] = s0, co[3] = lead;          // UniPoly::from_evals_deg3 (univariate.rs:103-113)
      co[2] = H::sub(H::mul(H::add(s1, sm1), tinv), s0);
      co[1] = H::sub(H::sub(H::sub(s1, lead), s0), co[2]);
    } else {
      const F s1 = H::sub(claim_i, t0);  // evals = [e0, claim - e0, bound coefficient]; from_evals_deg2 (univariate.rs:90-99)
      co[0] = t0, co[2] = t1;
      co[1] = H::sub(H::sub(s1, t1), t0);
    }
    const F r = h.ask(cb, cb_ctx, co, NCO, out_polys ? out_polys + 32 * NCO * (size_t)(j - 1) : nullptr, out_r ? out_r + 32 * (size_t)(j - 1) : nullptr);
    claim_i = H::poly_eval(co, NCO, r);
    if (MODE == 3) eq.bound(r);
    h.prof.rounds++;
    if (j < l) {
      if (MODE == 3) tb = eq.tables(j + 1);
      seq = pass.bind_sums(len, r, tb);
      len /= 2;
    } else {
      uint32_t* X[3] = {pass.A, pass.B, pass.C};
      seq = sc_final<FID>(h, X, NT, r, 0);
      const uint32_t* fin = h.wait(0, seq);
      if (out_claims)
        for (uint32_t i = 0; i < NT; i++) memcpy(out_claims + 32 * i, fin + 8 * i, 32);  // stored elements: already the vectors' form
    }
  }
  if (l == 0 && out_claims) {  // no rounds: the tables are their own evaluations
    stream_wait(c.stream);
    uint32_t* X[3] = {pass.A, pass.B, pass.C};
    for (uint32_t i = 0; i < NT; i++) HIPCHK(hipMemcpy(out_claims + 32 * i, X[i], 32, hipMemcpyDeviceToHost));
  }
  stream_wait(c.stream);  // the bound tables are the caller's again
  h.finish_profile(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - T0).count());
}

// SumcheckProof::prove_batch_eval (sumcheck.rs:251-353): k claims P_i(x_i) = e_i over polynomials of different sizes; the
// polynomials are bound in place (the reference binds clones, spartan/mod.rs:407-410: hand in copies to keep the originals).
template <int FID>
static void sc_prove_batch_t(Ctx& c, const uint8_t* claims, const size_t* num_rounds, void* const* polys, const uint8_t* const* eq_points,
                             const uint8_t* coeffs, size_t k, uint32_t flags, TranscriptFn cb, void* cb_ctx, uint8_t* out_polys,
                             uint8_t* out_r, uint8_t* out_finals) {
  using F = Fp<FID>;
  using H = ScHost<FID>;
  const auto T0 = std::chrono::steady_clock::now();
  require(k >= 1 && k <= kMailSlots, NMX_E_ARG, "prove_batch_eval: between 1 and 16 claims");
  H h(c, flags);
  uint32_t nmax = 0;
  size_t heap_total = 0;
  for (size_t i = 0; i < k; i++) {
    require(num_rounds[i] >= 1 && num_rounds[i] < 31, NMX_E_ARG, "prove_batch_eval: 1 <= num_rounds < 31");
    nmax = num_rounds[i] > nmax ? (uint32_t)num_rounds[i] : nmax;
    heap_total += (ScEq<FID>::heap_bytes((uint32_t)num_rounds[i]) + 255) & ~(size_t)255;
  }
  arena_reserve(c, k * kScPartialBytes + 512);
  aux_reserve(c, heap_total);
  std::vector<ScEq<FID>> eq(k);
  std::vector<ScPass<FID, 1>> pass;
  pass.reserve(k);
  std::vector<F> cl(k), run(k), co(k);
  std::vector<size_t> len(k);
  std::vector<uint32_t> seq(k, 0);
  std::vector<typename ScEq<FID>::Tables> tb(k);
  size_t off = 0;
  for (size_t i = 0; i < k; i++) {
    eq[i].init(h, eq_points[i], (uint32_t)num_rounds[i], c.aux + off, flags);
    off += (ScEq<FID>::heap_bytes((uint32_t)num_rounds[i]) + 255) & ~(size_t)255;
    pass.emplace_back(h, polys[i], nullptr, nullptr, (uint32_t*)(c.arena + i * kScPartialBytes), (uint32_t)i);
    cl[i] = run[i] = h.in(claims + 32 * i);
    co[i] = h.in(coeffs + 32 * i);
    len[i] = (size_t)1 << num_rounds[i];
  }
  F e = F::zero();  // (:281-289) e = sum claim_i 2^(nmax - n_i) coeff_i
  for (size_t i = 0; i < k; i++) e = H::add(e, H::mul(H::mul(cl[i], h.pow2(nmax - (uint32_t)num_rounds[i])), co[i]));
  const F tinv = h.two_inv();
  // the first sums of every polynomial that starts in round 0
  for (size_t i = 0; i < k; i++)
    if (num_rounds[i] == nmax) {
      tb[i] = eq[i].tables(1);
      seq[i] = pass[i].sums(pass[i].A, nullptr, nullptr, len[i], tb[i]);
    }
  std::vector<F> e0(k), em1(k);
  for (uint32_t round = 0; round < nmax; round++) {
    const uint32_t remaining = nmax - round;
    for (size_t i = 0; i < k; i++) {
      if (remaining <= num_rounds[i]) {  // (:301-305)
        const F t0 = h.raw(h.wait((uint32_t)i, seq[i]), pass[i].factors(tb[i]));
        F lead;
        eq[i].derive(t0, F::zero(), run[i], true, e0[i], lead, em1[i], [&] { return pass[i].high_half_sum(len[i], tb[i]); });
      } else {  // not yet started: constant (:306-312)
        e0[i] = em1[i] = H::mul(h.pow2(remaining - (uint32_t)num_rounds[i] - 1), cl[i]);
      }
    }
    F c0 = F::zero(), cm1 = F::zero();
    for (size_t i = 0; i < k; i++) c0 = H::add(c0, H::mul(e0[i], co[i])), cm1 = H::add(cm1, H::mul(em1[i], co[i]));
    const F c1 = H::sub(e, c0);
    const F qc = H::mul(H::sub(H::add(c1, cm1), H::dbl(c0)), tinv);  // (S(1) + S(-1) - 2 S(0)) / 2
    F poly[3] = {c0, H::sub(H::sub(c1, qc), c0), qc};               // from_evals_deg2([S(0), S(1), quad])
    const F r = h.ask(cb, cb_ctx, poly, 3, out_polys ? out_polys + 96 * (size_t)round : nullptr, out_r ? out_r + 32 * (size_t)round : nullptr);
    h.prof.rounds++;
    for (size_t i = 0; i < k; i++) {
      if (remaining <= num_rounds[i]) {
        // update_claim (:68-75) with evals [e0, 0, em1]: a1 = (e1 - em1)/2, a2 = (e1 + em1)/2 - e0, claim' = e0 + r (a1 + r a2)
        const F e1 = H::sub(run[i], e0[i]);
        const F a1 = H::mul(H::sub(e1, em1[i]), tinv), a2 = H::sub(H::mul(H::add(e1, em1[i]), tinv), e0[i]);
        run[i] = H::add(e0[i], H::mul(r, H::add(a1, H::mul(r, a2))));
        eq[i].bound(r);
        if (len[i] > 2) {
          tb[i] = eq[i].tables(eq[i].round);
          seq[i] = pass[i].bind_sums(len[i], r, tb[i]);
        } else {
          uint32_t* X[3] = {pass[i].A, nullptr, nullptr};
          seq[i] = sc_final<FID>(h, X, 1, r, (uint32_t)i);
        }
        len[i] /= 2;
      } else if (remaining - 1 == num_rounds[i]) {  // joins in the next round: its first sums, unbound
        tb[i] = eq[i].tables(1);
        seq[i] = pass[i].sums(pass[i].A, nullptr, nullptr, len[i], tb[i]);
      }
    }
    e = H::poly_eval(poly, 3, r);
  }
  for (size_t i = 0; i < k; i++) {  // poly_finals (:347-349)
    const uint32_t* fin = h.wait((uint32_t)i, seq[i]);
    if (out_finals) memcpy(out_finals + 32 * i, fin, 32);
  }
  stream_wait(c.stream);
  h.finish_profile(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - T0).count());
}

void fv_sumcheck_prove(Ctx& c, int field, int which, const void* claim, const void* taus, size_t num_rounds, void* A, void* B, void* C,
                       uint32_t flags, TranscriptFn cb, void* cb_ctx, uint8_t* out_polys, uint8_t* out_r, uint8_t* out_claims) {
#define SCP(FID)                                                                                                              \
  if (which == 3) sc_prove_t<FID, 3>(c, claim, taus, num_rounds, A, B, C, flags, cb, cb_ctx, out_polys, out_r, out_claims);   \
  else sc_prove_t<FID, 4>(c, claim, taus, num_rounds, A, B, C, flags, cb, cb_ctx, out_polys, out_r, out_claims);              \
  return;
  switch (field) {
    case 0: SCP(0)
    case 1: SCP(1)
    case 2: SCP(2)
    case 3: SCP(3)
    default: throw Fail{NMX_E_ARG, "bad field id"};
  }
#undef SCP
}
void fv_sumcheck_prove_batch(Ctx& c, int field, const uint8_t* claims, const size_t* num_rounds, void* const* polys,
                             const uint8_t* const* eq_points, const uint8_t* coeffs, size_t k, uint32_t flags, TranscriptFn cb, void* cb_ctx,
                             uint8_t* out_polys, uint8_t* out_r, uint8_t* out_finals) {
  switch (field) {
    case 0: sc_prove_batch_t<0>(c, claims, num_rounds, polys, eq_points, coeffs, k, flags, cb, cb_ctx, out_polys, out_r, out_finals); return;
    case 1: sc_prove_batch_t<1>(c, claims, num_rounds, polys, eq_points, coeffs, k, flags, cb, cb_ctx, out_polys, out_r, out_finals); return;
    case 2: sc_prove_batch_t<2>(c, claims, num_rounds, polys, eq_points, coeffs, k, flags, cb, cb_ctx, out_polys, out_r, out_finals); return;
    case 3: sc_prove_batch_t<3>(c, claims, num_rounds, polys, eq_points, coeffs, k, flags, cb, cb_ctx, out_polys, out_r, out_finals); return;
    default: throw Fail{NMX_E_ARG, "bad field id"};
  }
}

}  // namespace nmx
