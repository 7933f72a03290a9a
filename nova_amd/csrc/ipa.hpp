// ipa.hpp -- the field side of the inner-product argument (InnerProductArgument::prove, /root/reference/src/provider/ipa_pc.rs:174-281;
// EvaluationEngine::prove :69-82 = the evaluation argument of S2 in CompressedSNARK::prove, src/nova/mod.rs:862-881), included by
// sumcheck.hip (it uses that file's block sums and challenge conversion, and the provers' pinned mailbox of sumcheck_prove.hpp).
//
// The reference folds the commitment key every round -- ck' = ck.fold(r^-1, r): n/2 two-point MSMs (pedersen.rs:484-497), 2n scalar
// multiplications per proof -- and commits the round's L and R against the folded halves.  Here the key is never folded.  With
// S_k[m] = prod_{t <= k} r_t^(+-1) (r_t when bit (k - t) of m is set, r_t^-1 otherwise; 2^k entries) the round-k key is
//     ck_k[i] = sum over the 2^k indices j = m * n_k + i of S_k[m] * ck[j]            (n_k = n / 2^k, i < n_k)
// so    L_k = <a_k[..h], ck_k[h..]> + c_L U  =  sum_j vL[j] * ck[j] + c_L U,   vL[m n_k + h + i] = a_k[i] * S_k[m],  0 elsewhere
//       R_k = <a_k[h..], ck_k[..h]> + c_R U  =  sum_j vR[j] * ck[j] + c_R U,   vR[m n_k + i]     = a_k[h + i] * S_k[m]   (h = n_k / 2)
// -- two MSMs over the ORIGINAL key (resident in HBM with its window tables, the same key every commitment of that curve uses) with n/2
// non-zero scalars each, run as one fused two-vector batch.  Same group elements, hence the same affine L, R, the same transcript and
// the same a_hat; 2 n log n products in the field replace 2 n scalar multiplications on the curve.
//   k_ipa_round    ONE launch per round: the folds of the previous round -- a' = a_L r + r^-1 a_R, b' = b_L r^-1 + r b_R (:237-247),
//                  S_k[2m] = S_{k-1}[m] r^-1, S_k[2m + 1] = S_{k-1}[m] r -- recomputed by every lane that needs a value (a lane that
//                  waited for a neighbour's store would need a grid-wide barrier) and stored once for the next round; vL, vR; and the
//                  block partials of c_L = <a_k[..h], b_k[h..]>, c_R = <a_k[h..], b_k[..h]> (:207-208), written straight into pinned
//                  host memory (the provers' mailbox), where a POOL THREAD adds them up once the kernel's event has fired and computes
//                  the blinding terms c_L ck_c, c_R ck_c while the calling thread is already enqueueing the round's MSM: one host
//                  synchronisation per round, at the MSM's end.
// Vectors are processed in the form they arrive in (canonical or R = 2^256 Montgomery: every map here is linear in a and in b
// separately); S is kept in the kernels' internal residue form.  HBM traffic per round: 2 x 32 n bytes written (vL, vR), read by the
// MSM's digit pass -- at 2^14..2^17 elements every launch here is latency, the round's time is the fused MSM's.
#pragma once

namespace nmx {

template <int FID> struct IpaRoundArgs {
  const uint32_t *a, *b;            // the previous round's vectors, 2 * len elements (FIRST: the caller's, len elements)
  const uint32_t* S;                // S_{k-1}: n / (2 len) entries (FIRST: unused, S_0 = [1])
  uint32_t *a_out, *b_out, *S_out;  // this round's vectors (len elements) and table (n / len entries) (FIRST: unused)
  uint32_t *vL, *vR;                // n elements each
  uint32_t* partial;                // 16 words per block, pinned host memory
  Fp<FID> r, rinv;                  // the previous round's challenge: r * 2^261, r^-1 * 2^261, canonical
  uint32_t n, len, shift;           // len = n_k, shift = log2 len
};

template <int FID, bool FIRST> __global__ __launch_bounds__(256) void k_ipa_round(IpaRoundArgs<FID> p) {
  using F = Fp<FID>;
  __shared__ uint32_t lds[72];
  const uint32_t len = p.len, h = len >> 1, stride = gridDim.x * 256u, g0 = blockIdx.x * 256u + threadIdx.x;
  const F z = F::zero();
  // element x of this round's a / b: the caller's, or the fold of the previous round's halves (one reduction for both products)
  auto a_at = [&](uint32_t x) { return FIRST ? ldw<FID>(p.a, x) : F::mul_add(ldw<FID>(p.a, x), p.r, ldw<FID>(p.a, (size_t)x + len), p.rinv).norm(); };
  auto b_at = [&](uint32_t x) { return FIRST ? ldw<FID>(p.b, x) : F::mul_add(ldw<FID>(p.b, x), p.rinv, ldw<FID>(p.b, (size_t)x + len), p.r).norm(); };
  auto s_at = [&](uint32_t m) { return FIRST ? F::one() : (ldw<FID>(p.S, m >> 1) * ((m & 1u) ? p.r : p.rinv)); };
  for (uint32_t j = g0; j < p.n; j += stride) {
    const uint32_t m = j >> p.shift, t = j & (len - 1u);
    const bool right = t >= h;
    const F x = (a_at(right ? t - h : t + h) * s_at(m)).canon();
    (right ? x : z).to_words(p.vL + 8 * (size_t)j);
    (right ? z : x).to_words(p.vR + 8 * (size_t)j);
    if (!FIRST && j < (p.n >> p.shift)) s_at(j).canon().to_words(p.S_out + 8 * (size_t)j);
  }
  F sL = F::zero(), sR = F::zero();
  uint32_t pending = 0;
  for (uint32_t i = g0; i < h; i += stride) {
    const F al = a_at(i), ar = a_at(i + h), bl = b_at(i), br = b_at(i + h);
    if (!FIRST) {
      al.canon().to_words(p.a_out + 8 * (size_t)i), ar.canon().to_words(p.a_out + 8 * ((size_t)i + h));
      bl.canon().to_words(p.b_out + 8 * (size_t)i), br.canon().to_words(p.b_out + 8 * ((size_t)i + h));
    }
    sL = sL + al * br;
    sR = sR + ar * bl;
    if (++pending == 6) {
      sL = sL.norm().canon();
      sR = sR.norm().canon();
      pending = 0;
    }
  }
  sL = sL.norm().canon();
  sR = sR.norm().canon();
  block_sum_pair<FID>(sL, sR, lds);
  if (threadIdx.x == 0) {
    sL.to_words(p.partial + 16 * blockIdx.x);
    sR.to_words(p.partial + 16 * blockIdx.x + 8);
  }
}

// the last fold: a_hat = a[0] r + r^-1 a[1] (:237-241 of the final round; :271)
template <int FID> struct IpaLastFn {
  const uint32_t* a;
  uint32_t* out;
  Fp<FID> r, rinv;
  NMX_HD void operator()(uint32_t) const {
    using F = Fp<FID>;
    F::mul_add(F::from_words(a), r, F::from_words(a + 8), rinv).norm().canon().to_words(out);
  }
};

static inline uint32_t ipa_blocks(size_t n) {
  const size_t want = (n + 255) / 256;
  return (uint32_t)(want < 1 ? 1 : want > 128 ? 128 : want);
}

// One round's launch.  r_prev / rinv_prev: the previous round's challenge in the ABI form (nullptr: the first round).  The kernel's
// completion is recorded in `done`; the partials land in the context's mailbox (host memory).
template <int FID>
static void ipa_round_t(Ctx& c, const uint32_t* a, const uint32_t* b, const uint32_t* S, uint32_t* a_out, uint32_t* b_out, uint32_t* S_out,
                        size_t n, size_t len, const void* r_prev, const void* rinv_prev, uint32_t flags, uint32_t* vL, uint32_t* vR,
                        hipEvent_t done, const uint32_t** partial_host) {
  const bool mont = flags & NMX_SCALARS_MONT;
  ScDev<FID> h(c, flags);  // (allocates the mailbox on first use)
  IpaRoundArgs<FID> p{};
  p.a = a, p.b = b, p.S = S, p.a_out = a_out, p.b_out = b_out, p.S_out = S_out, p.vL = vL, p.vR = vR, p.partial = h.part_dev(0);
  *partial_host = h.part_host(0);
  p.n = (uint32_t)n, p.len = (uint32_t)len;
  while (((size_t)1 << p.shift) < len) p.shift++;
  const uint32_t blocks = ipa_blocks(n);
  static_assert(128 * 64 <= kMailSlots * kPartSlotBytes, "the partials of a round fit the mailbox's partial area");
  if (r_prev) {
    p.r = challenge_internal<FID>(r_prev, mont), p.rinv = challenge_internal<FID>(rinv_prev, mont);
    hipLaunchKernelGGL((k_ipa_round<FID, false>), dim3(blocks), dim3(256), 0, c.stream, p);
  } else {
    hipLaunchKernelGGL((k_ipa_round<FID, true>), dim3(blocks), dim3(256), 0, c.stream, p);
  }
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(done, c.stream));
}

// c_L (which = 0) or c_R (1) of the round whose kernel has completed, in the ABI form of `flags`: the host adds the block partials
// (a sum of products of two stored elements each: ScDev::raw with k = 2)
template <int FID> static void ipa_scalar_t(const uint32_t* area, size_t n, int which, uint32_t flags, uint8_t* out32) {
  using H = HostFp4<FID>;
  static const std::array<H, 5> tab[2] = {ScDev<FID>::make_corr(false), ScDev<FID>::make_corr(true)};
  const bool mont = flags & NMX_SCALARS_MONT;
  H acc = H::zero();
  const uint32_t blocks = ipa_blocks(n);
  for (uint32_t i = 0; i < blocks; i++) acc = acc + H::from_plain_times(area + 16 * (size_t)i + 8 * which, tab[mont ? 1 : 0][2]);
  if (mont) acc.to_mont256(out32);
  else acc.to_canonical(out32);
}

template <int FID> static void ipa_last_t(Ctx& c, const uint32_t* a, const void* r, const void* rinv, uint32_t flags, uint32_t* dout, uint8_t* out32) {
  const bool mont = flags & NMX_SCALARS_MONT;
  IpaLastFn<FID> f{a, dout, challenge_internal<FID>(r, mont), challenge_internal<FID>(rinv, mont)};
  DeviceBackend be(c, false, false);
  be.launch(f, 1);
  be.d2h(out32, dout, 32);
  be.sync();
}

// S_0 = [1] in the internal form
template <int FID> static void ipa_one_t(Ctx& c, uint32_t* S) {
  uint32_t w[8];
  Fp<FID>::one().canon().to_words(w);
  HIPCHK(hipMemcpyAsync(S, w, 32, hipMemcpyHostToDevice, c.stream));
  HIPCHK(hipStreamSynchronize(c.stream));  // `w` is a stack word
}

// 1 / r in the ABI form of `flags` (false: r = 0 -- `r.invert().unwrap()`, ipa_pc.rs:235, panics there)
template <int FID> static bool ipa_invert_t(const void* r, uint32_t flags, void* out) {
  using H = HostFp4<FID>;
  uint32_t w[8];
  memcpy(w, r, 32);
  require(Fp<FID>::words_lt_p(w), NMX_E_SCALAR_RANGE, "challenge >= field modulus");
  const bool mont = flags & NMX_SCALARS_MONT;
  const H x = mont ? H::from_mont256(r) : H::from_canonical(r);
  if (x.is_zero()) return false;
  const H y = x.inv();
  if (mont) y.to_mont256(out);
  else y.to_canonical(out);
  return true;
}

#define NMX_IPA_FIELD_SWITCH(CALL)                      \
  switch (field) {                                      \
    case 0: CALL(0); break;                             \
    case 1: CALL(1); break;                             \
    case 2: CALL(2); break;                             \
    case 3: CALL(3); break;                             \
    default: throw Fail{NMX_E_ARG, "bad field id"};     \
  }

void fv_ipa_round(Ctx& c, int field, const uint32_t* a, const uint32_t* b, const uint32_t* S, uint32_t* a_out, uint32_t* b_out, uint32_t* S_out,
                  size_t n, size_t len, const void* r_prev, const void* rinv_prev, uint32_t flags, uint32_t* vL, uint32_t* vR, hipEvent_t done,
                  const uint32_t** partial_host) {
#define X(FID) ipa_round_t<FID>(c, a, b, S, a_out, b_out, S_out, n, len, r_prev, rinv_prev, flags, vL, vR, done, partial_host)
  NMX_IPA_FIELD_SWITCH(X)
#undef X
}
void fv_ipa_scalar(int field, const uint32_t* partial_host, size_t n, int which, uint32_t flags, uint8_t* out32) {
#define X(FID) ipa_scalar_t<FID>(partial_host, n, which, flags, out32)
  NMX_IPA_FIELD_SWITCH(X)
#undef X
}
void fv_ipa_last(Ctx& c, int field, const uint32_t* a, const void* r, const void* rinv, uint32_t flags, uint32_t* dout, uint8_t* out32) {
#define X(FID) ipa_last_t<FID>(c, a, r, rinv, flags, dout, out32)
  NMX_IPA_FIELD_SWITCH(X)
#undef X
}
void fv_ipa_one(Ctx& c, int field, uint32_t* S) {
#define X(FID) ipa_one_t<FID>(c, S)
  NMX_IPA_FIELD_SWITCH(X)
#undef X
}
bool fv_ipa_invert(int field, const void* r, uint32_t flags, void* out) {
  bool ok = false;
#define X(FID) ok = ipa_invert_t<FID>(r, flags, out)
  NMX_IPA_FIELD_SWITCH(X)
#undef X
  return ok;
}
#undef NMX_IPA_FIELD_SWITCH

}  // namespace nmx
