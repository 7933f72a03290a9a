// ipa.hpp -- the field side of the inner-product argument (InnerProductArgument::prove, /root/reference/src/provider/ipa_pc.rs:174-281;
// EvaluationEngine::prove :69-82 = the evaluation argument of S2 in CompressedSNARK::prove, src/nova/mod.rs:862-881), included by
// sumcheck.hip (it uses that file's block sums and the partial-sum launch).
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
//   k_ipa_expand   vL, vR from (a_k, S_k), and the block partials of c_L = <a_k[..h], b_k[h..]>, c_R = <a_k[h..], b_k[..h]>  (:207-208)
//   k_ipa_fold     a' = a_L r + r^-1 a_R, b' = b_L r^-1 + r b_R (:237-247) and S_{k+1}[2m] = S_k[m] r^-1, S_{k+1}[2m + 1] = S_k[m] r
// Vectors are processed in the form they arrive in (canonical or R = 2^256 Montgomery: every map here is linear in a and in b
// separately); S is kept in the kernels' internal residue form.  HBM traffic per round: 2 x 32 n bytes written (vL, vR), read by the
// MSM's digit pass -- at 2^14..2^17 elements every launch here is latency, the round's time is the fused MSM's.
#pragma once

namespace nmx {

template <int FID>
__global__ __launch_bounds__(256) void k_ipa_expand(const uint32_t* a, const uint32_t* b, const uint32_t* S, uint32_t n, uint32_t len,
                                                    uint32_t shift /* log2 len */, uint32_t* vL, uint32_t* vR,
                                                    uint32_t* partial /* 16 words per block */) {
  using F = Fp<FID>;
  __shared__ uint32_t lds[72];
  const uint32_t h = len >> 1, stride = gridDim.x * 256u;
  const F z = F::zero();
  for (uint32_t j = blockIdx.x * 256u + threadIdx.x; j < n; j += stride) {
    const uint32_t m = j >> shift, t = j & (len - 1u);
    const F s = ldw<FID>(S, m);
    const bool right = t >= h;
    const F x = (ldw<FID>(a, right ? t - h : t + h) * s).canon();
    (right ? x : z).to_words(vL + 8 * (size_t)j);
    (right ? z : x).to_words(vR + 8 * (size_t)j);
  }
  F sL = F::zero(), sR = F::zero();
  uint32_t pending = 0;
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < h; i += stride) {
    sL = sL + ldw<FID>(a, i) * ldw<FID>(b, (size_t)i + h);
    sR = sR + ldw<FID>(a, (size_t)i + h) * ldw<FID>(b, i);
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
    sL.to_words(partial + 16 * blockIdx.x);
    sR.to_words(partial + 16 * blockIdx.x + 8);
  }
}

template <int FID> struct IpaFoldFn {
  const uint32_t *a, *b, *S;
  uint32_t *a_out, *b_out, *S_out;
  Fp<FID> r, rinv;  // r * 2^261, r^-1 * 2^261, canonical
  uint32_t h, s_len;
  NMX_HD void operator()(uint32_t i) const {
    using F = Fp<FID>;
    if (i < h) {
      const F al = F::from_words(a + 8 * (size_t)i), ar = F::from_words(a + 8 * ((size_t)i + h));
      const F bl = F::from_words(b + 8 * (size_t)i), br = F::from_words(b + 8 * ((size_t)i + h));
      F::mul_add(al, r, ar, rinv).norm().canon().to_words(a_out + 8 * (size_t)i);  // one reduction for both products
      F::mul_add(bl, rinv, br, r).norm().canon().to_words(b_out + 8 * (size_t)i);
    }
    if (i < s_len) {
      const F s = F::from_words(S + 8 * (size_t)i);
      (s * rinv).canon().to_words(S_out + 8 * (size_t)(2 * i));
      (s * r).canon().to_words(S_out + 8 * (size_t)(2 * i + 1));
    }
  }
};

static inline uint32_t ipa_blocks(size_t n) {
  const size_t want = (n + 256 * 4 - 1) / (256 * 4);
  return (uint32_t)(want < 1 ? 1 : want > 1024 ? 1024 : want);
}

template <int FID>
static void ipa_expand_t(Ctx& c, const uint32_t* a, const uint32_t* b, const uint32_t* S, size_t n, size_t len, uint32_t flags,
                         uint32_t* vL, uint32_t* vR, uint32_t* partial, uint32_t* dout, uint8_t* out_c64) {
  using F = Fp<FID>;
  const bool mont = flags & NMX_SCALARS_MONT;
  uint32_t shift = 0;
  while (((size_t)1 << shift) < len) shift++;
  const uint32_t blocks = ipa_blocks(n);
  DeviceBackend be(c, false, false);
  hipLaunchKernelGGL((k_ipa_expand<FID>), dim3(blocks), dim3(256), 0, c.stream, a, b, S, (uint32_t)n, (uint32_t)len, shift, vL, vR, partial);
  HIPCHK(hipGetLastError());
  hipLaunchKernelGGL((k_sum_partials_n<FID, 2, 2>), dim3(1), dim3(256), 0, c.stream, partial, blocks, dout);
  HIPCHK(hipGetLastError());
  uint32_t res[16];
  be.d2h(res, dout, 64);  // through the context's pinned landing buffer
  be.sync();
  // a product of two stored elements is x * Fm^2 / R': back to the vectors' own form x * Fm by R'^2 / Fm (one more product)
  const F corr = pow2_plain<FID>(261u * 2 - (mont ? 256u : 0u));
  for (int j = 0; j < 2; j++) {
    const F v = F::from_words(res + 8 * j) * corr;
    uint32_t w[8];
    v.canon().to_words(w);
    memcpy(out_c64 + 32 * j, w, 32);
  }
}

template <int FID>
static void ipa_fold_t(Ctx& c, const uint32_t* a, const uint32_t* b, size_t len, const void* r, const void* rinv, uint32_t flags,
                       uint32_t* a_out, uint32_t* b_out, const uint32_t* S, size_t s_len, uint32_t* S_out) {
  const bool mont = flags & NMX_SCALARS_MONT;
  IpaFoldFn<FID> f{a, b, S, a_out, b_out, S_out, challenge_internal<FID>(r, mont), challenge_internal<FID>(rinv, mont),
                   (uint32_t)(len / 2), (uint32_t)s_len};
  DeviceBackend be(c, false, false);
  be.launch(f, (uint32_t)(len / 2 > s_len ? len / 2 : s_len));
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

void fv_ipa_expand(Ctx& c, int field, const uint32_t* a, const uint32_t* b, const uint32_t* S, size_t n, size_t len, uint32_t flags,
                   uint32_t* vL, uint32_t* vR, uint32_t* partial, uint32_t* dout, uint8_t* out_c64) {
#define X(FID) ipa_expand_t<FID>(c, a, b, S, n, len, flags, vL, vR, partial, dout, out_c64)
  NMX_IPA_FIELD_SWITCH(X)
#undef X
}
void fv_ipa_fold(Ctx& c, int field, const uint32_t* a, const uint32_t* b, size_t len, const void* r, const void* rinv, uint32_t flags,
                 uint32_t* a_out, uint32_t* b_out, const uint32_t* S, size_t s_len, uint32_t* S_out) {
#define X(FID) ipa_fold_t<FID>(c, a, b, len, r, rinv, flags, a_out, b_out, S, s_len, S_out)
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
size_t fv_ipa_partial_bytes(size_t n) { return (size_t)ipa_blocks(n) * 64; }
#undef NMX_IPA_FIELD_SWITCH

}  // namespace nmx
