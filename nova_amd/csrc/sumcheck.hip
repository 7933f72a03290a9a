// sumcheck.hip -- the N-scaling sums of Spartan's eq-factored sum-check rounds.
//
// One launch computes, over id in [0, h) with (x0, x1) = (X[id], X[id + h]) and
// factor(id) = eqL[id >> shift] * eqR[id & mask]   (first-half rounds; eqL == nullptr: factor = eqR[id], last half),
//   mode 3  t_0 = sum (a0*b0 - c0) * factor,  t_inf = sum (a1-a0)*(b1-b0) * factor
//           EqSumCheckInstance::evaluation_points_cubic_with_three_inputs, /root/reference/src/spartan/sumcheck.rs:900-958
//   mode 2  t_0 = sum (a0*b0 - 1) * factor,   t_inf as above      ..._cubic_with_two_inputs, sumcheck.rs:972-1037
//   mode 1  t_0 = sum a0 * factor                                 ..._quadratic_with_one_input, sumcheck.rs:1039-1075
// The O(1) derivation of the round polynomial from (t_0, t_inf, claim) (sumcheck.rs:686-753), the eq tables
// (O(sqrt n), sumcheck.rs:608-660) and the transcript stay on the host side of the reference.
//
// HBM-bound: 160 B (mode 3) / 128 B (mode 2) / 32 B (mode 1) per index against 6 / 5 / 2 modmuls.  Lanes read
// consecutive elements (32 B per lane, coalesced); per-lane partial sums are combined by an LDS tree per block and a
// second one-block launch.  Vectors are processed in the form they arrive in (canonical or R = 2^256 Montgomery):
// every term carries the same power of the form factor, fixed by one constant multiplication on the host.
#include "runtime.hpp"
#include "host_fp4.hpp"

namespace nmx {

template <int FID> __device__ __forceinline__ Fp<FID> ldw(const uint32_t* p, size_t i) {
  return Fp<FID>::from_words(p + 8 * i);
}

// tree-reduce `v` over the 256 lanes of a block; result valid in lane 0.  Values < 2p on entry.
template <int FID> __device__ __forceinline__ Fp<FID> block_sum(Fp<FID> v, uint32_t* lds /* 256 x 9 */) {
  const uint32_t t = threadIdx.x;
#pragma unroll
  for (int i = 0; i < 9; i++) lds[i * 256 + t] = v.l[i];  // limb-major: conflict-free
  __syncthreads();
  for (uint32_t s = 128; s >= 1; s >>= 1) {
    if (t < s) {
      Fp<FID> o;
#pragma unroll
      for (int i = 0; i < 9; i++) o.l[i] = lds[i * 256 + t + s];
      v = (v + o).norm().canon();  // < 4p -> < p
#pragma unroll
      for (int i = 0; i < 9; i++) lds[i * 256 + t] = v.l[i];
    }
    __syncthreads();
  }
  return v;
}

// Both sums of a pass at once, the first six levels inside the wave: shuffles instead of LDS round trips and barriers, limbs
// normalised every second step (inputs canonical), a wave's sum < 64 p brought back below 2 p by one product with 1; then the
// four wave sums through LDS.  Result valid in thread 0.  The two eight-level LDS trees it replaces in k_eq_rows were ~10 % of a
// block's time at two rows per block.
// LADDER: the wave sum comes down by six conditional subtractions instead of a product with ONE (Fp::canon_below) -- for the one-block
// and few-block passes of the provers, whose cost is their dependent chain; the streaming passes keep the product (fewer instructions)
template <int FID, int J, bool LADDER = false>
__device__ __forceinline__ void block_sum_waves(Fp<FID> (&x)[J], uint32_t* lds /* >= 36 J words */) {
  using F = Fp<FID>;
  const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
#pragma unroll
  for (int j = 0; j < J; j++) {
#pragma unroll
    for (uint32_t d = 32; d >= 1; d >>= 1) {
#pragma unroll
      for (int i = 0; i < 9; i++) x[j].l[i] += (uint32_t)__shfl_down((int)x[j].l[i], d, 64);
      if (d == 16 || d == 4 || d == 1) x[j] = x[j].norm();
    }
    x[j] = LADDER ? x[j].template canon_below<5>() : (x[j] * F::one()).canon4();  // lane 0: the wave's sum, < p
    if (lane == 0) {
#pragma unroll
      for (int i = 0; i < 9; i++) lds[(j * 9 + i) * 4 + wave] = x[j].l[i];
    }
  }
  __syncthreads();
  if (t == 0) {
#pragma unroll
    for (int j = 0; j < J; j++) {
      F acc = F::zero();
#pragma unroll
      for (int w = 0; w < 4; w++) {
        F o;
#pragma unroll
        for (int i = 0; i < 9; i++) o.l[i] = lds[(j * 9 + i) * 4 + w];
        acc = acc + o;
      }
      x[j] = acc.norm().canon();  // < 4 p -> < p
    }
  }
}
template <int FID, bool LADDER = false>
__device__ __forceinline__ void block_sum_pair(Fp<FID>& g0, Fp<FID>& g1, uint32_t* lds /* >= 72 words */) {
  Fp<FID> x[2] = {g0, g1};
  block_sum_waves<FID, 2, LADDER>(x, lds);
  if (threadIdx.x == 0) {
    g0 = x[0];
    g1 = x[1];
  }
}

// Final sums of the per-block partials: one block behind the pass.  Round 4 measured the alternative -- the block that draws the
// last of gridDim tickets adds the partials up inside the pass itself (partials published as agent-scope relaxed atomics after
// s_waitcnt vmcnt(0), one acquire fence in the last block) -- on one box against this form: sumcheck3 2^24 0.419 / 0.389 ms
// against 0.422 / 0.422, round3 0.617 / 0.623 against 0.637 / 0.626, but quad_prod 0.226 / 0.231 against 0.214 / 0.225 and a
// 2^20 evaluation 29.9 / 29.5 us against 27.4 / 27.8 (profiles/r04_fieldvec/final_sum_ab.txt): a wash -- the dependent launch
// costs 2-3 us of gap, the ticket + fence + the same serial tail cost as much -- so the simpler form, whose ordering is a
// kernel boundary, stays.  (With agent-scope acq_rel on the ticket in EVERY block the passes were 8-20 % slower: a release /
// acquire pair is a write-back + invalidate of the whole L2.)  The sum itself now runs six levels in shuffles and one LDS hop
// (block_sum_waves) instead of an eight-level LDS tree.
template <int FID, int J, int STRIDE> __global__ __launch_bounds__(256) void k_sum_partials_n(const uint32_t* partial, uint32_t nparts, uint32_t* out) {
  using F = Fp<FID>;
  __shared__ uint32_t lds[36 * J];
  F s[J];
#pragma unroll
  for (int j = 0; j < J; j++) s[j] = F::zero();
  // This launch is pure latency behind the pass (7-21 us for 512-2048 partials under rocprofv3).  Issuing all of a thread's
  // loads before the first use (eight 16-byte pairs in flight, 208 registers) was measured against this loop on one box and is no
  // faster (mle_eval 2^20 29.1 against 28.2 us, sumcheck3 2^24 0.414-0.431 against 0.413-0.436 ms): the loop stays.
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
    for (int j = 0; j < J; j++) s[j].to_words(out + 8 * j);
  }
}

// The terms of one index: e0 = a0*b0 - c0 (at product scale) and q = (a1-a0)*(b1-b0), with the eq factor.
template <int FID, int MODE> struct EqTerm {
  Fp<FID> e0, q, fac;
};
template <int FID, int MODE>
__device__ __forceinline__ EqTerm<FID, MODE> eq_term(const uint32_t* A, const uint32_t* B, const uint32_t* C,
                                                     const uint32_t* eqL, const uint32_t* eqR, uint32_t shift,
                                                     uint32_t mask, uint32_t h, const Fp<FID>& nk, uint32_t id,
                                                     uint32_t eq_index = 0xffffffffu) {
  using F = Fp<FID>;
  EqTerm<FID, MODE> t;
  // eq_index given (k_eq_rows): the factor is eqR[eq_index] alone, eqL is applied once per row by the caller
  t.fac = ldw<FID>(eqR, eq_index != 0xffffffffu ? eq_index : (eqL ? (id & mask) : id));
  if (eqL) t.fac = ldw<FID>(eqL, id >> shift) * t.fac;
  const F a0 = ldw<FID>(A, id);
  if (MODE == 1) {
    t.e0 = a0;
    t.q = F::zero();
  } else {
    const F a1 = ldw<FID>(A, (size_t)id + h), b0 = ldw<FID>(B, id), b1 = ldw<FID>(B, (size_t)id + h);
    // a0*b0 - c0*k in ONE reduction: nk = p - k (k brings c0 to the scale of a product; MODE 2: c0 = 1 folded into nk)
    t.e0 = MODE == 3 ? F::mul_add(a0, b0, ldw<FID>(C, id), nk) : (a0 * b0 + nk).norm();   // < 1.02 p / < 2.01 p
    t.q = F::sub2(a1, a0).norm() * F::sub2(b1, b0).norm();                                 // operands < 3 p
  }
  return t;
}

template <int FID, int MODE>
__global__ __launch_bounds__(256) void k_eq_sums(const uint32_t* A, const uint32_t* B, const uint32_t* C,
                                                 const uint32_t* eqL, const uint32_t* eqR, uint32_t shift, uint32_t mask,
                                                 uint32_t h, Fp<FID> nk, uint32_t* partial) {
  using F = Fp<FID>;
  __shared__ uint32_t lds[9 * 256];
  F s0 = F::zero(), s1 = F::zero();
  uint32_t pending = 0;  // lazily added terms since the last canonicalisation (each < 1.1 p, limbs < 2^29)
  const uint32_t stride = gridDim.x * 256u;
  uint32_t id = blockIdx.x * 256u + threadIdx.x;
  // two indices per iteration: s += e_i * f_i + e_j * f_j is one reduction for two products
  for (; id + stride < h; id += 2 * stride) {
    const EqTerm<FID, MODE> x = eq_term<FID, MODE>(A, B, C, eqL, eqR, shift, mask, h, nk, id);
    const EqTerm<FID, MODE> y = eq_term<FID, MODE>(A, B, C, eqL, eqR, shift, mask, h, nk, id + stride);
    s0 = s0 + F::mul_add(x.e0, x.fac, y.e0, y.fac);            // < p (1 + 2 * 2.1 * 1.1 / 127)
    if (MODE != 1) s1 = s1 + F::mul_add(x.q, x.fac, y.q, y.fac);
    if (++pending == 6) {  // 1 canonical + 6 fresh terms: value < 8 p, limbs < 7 * 2^29 -- then back to < p
      s0 = s0.norm().canon();
      s1 = s1.norm().canon();
      pending = 0;
    }
  }
  if (id < h) {
    const EqTerm<FID, MODE> x = eq_term<FID, MODE>(A, B, C, eqL, eqR, shift, mask, h, nk, id);
    s0 = s0 + x.e0 * x.fac;
    if (MODE != 1) s1 = s1 + x.q * x.fac;
  }
  s0 = s0.norm().canon();
  s1 = s1.norm().canon();
  s0 = block_sum<FID>(s0, lds);
  if (MODE != 1) {
    __syncthreads();
    s1 = block_sum<FID>(s1, lds);
  }
  if (threadIdx.x == 0) {
    s0.to_words(partial + 16 * blockIdx.x);
    s1.to_words(partial + 16 * blockIdx.x + 8);
  }
}

// First-half rounds (both eq tables), factored by rows:  sum_hi eqL[hi] * (sum_lo term[hi, lo] * eqR[lo]).  The generic
// kernel pays eqL * eqR and term * factor per element -- 1.75 reductions per element in mode 1
// (MultilinearPolynomial::evaluate_with, multilinear.rs:98-129: multiplier-bound, 0.28 ms for 2^24 elements = 24 % of the HBM
// roofline), 5 per index in mode 3.  Here eqL enters once per lane and row and two terms share a reduction -- four in mode 1
// since round 3 (Fp::dot): 1 + 0.25 + 1/K reductions per element in mode 1, 4 + 2/K per index in mode 3.  A block walks whole rows (2^shift
// consecutive indices, coalesced); rows shorter than the block share it.
#ifndef NMX_EQROWS_MINWAVES
#define NMX_EQROWS_MINWAVES 1  // A/B knob: 4 forces 128 registers (modes 2 / 3 then spill 16)
#endif
template <int FID, int MODE>
__global__ __launch_bounds__(256, NMX_EQROWS_MINWAVES) void k_eq_rows(const uint32_t* A, const uint32_t* B, const uint32_t* C, const uint32_t* eqL,
                                                 const uint32_t* eqR, uint32_t shift, uint32_t h, Fp<FID> nk,
                                                 uint32_t* partial) {
  using F = Fp<FID>;
  __shared__ uint32_t lds[9 * 256];
  const uint32_t row = 1u << shift, P = row < 256u ? row : 256u, rows_per_block = 256u / P, K = row / P;
  const uint32_t t = threadIdx.x, lo0 = t % P, sub = t / P;
  const uint32_t nrows = (uint32_t)(((uint64_t)h + row - 1) >> shift);
  F g0 = F::zero(), g1 = F::zero();
  uint32_t pend_g = 0;
  for (uint32_t hi = blockIdx.x * rows_per_block + sub; hi < nrows; hi += gridDim.x * rows_per_block) {
    const uint32_t base = hi << shift;
    F r0 = F::zero(), r1 = F::zero();
    uint32_t pend = 0, k = 0;
    // mode 1 (evaluate_with): four indices per reduction (Fp::dot), whole groups inside the vector.  (Modes 2 / 3 would hold
    // twelve operands -- 221 registers, two waves per SIMD -- for 6 % fewer multiply-adds: they keep the pairs below.)
    if constexpr (MODE == 1) {
      for (; k + 3 < K; k += 4) {
        const uint32_t l0 = lo0 + k * P;
        if (base + l0 + 3 * P >= h) break;
        F e[4], fc[4];
#pragma unroll
        for (uint32_t j = 0; j < 4; j++) {
          e[j] = ldw<FID>(A, base + l0 + j * P);
          fc[j] = ldw<FID>(eqR, l0 + j * P);
        }
        r0 = r0 + F::template dot<4>(e, fc);  // < 1.04 p
        if (++pend == 6) {
          r0 = r0.norm().canon();
          pend = 0;
        }
      }
    }
    for (; k + 1 < K; k += 2) {  // two indices per reduction
      const uint32_t l0 = lo0 + k * P, l1 = l0 + P;
      if (base + l1 < h) {
        const EqTerm<FID, MODE> x = eq_term<FID, MODE>(A, B, C, nullptr, eqR, 0, 0, h, nk, base + l0, l0);
        const EqTerm<FID, MODE> y = eq_term<FID, MODE>(A, B, C, nullptr, eqR, 0, 0, h, nk, base + l1, l1);
        r0 = r0 + F::mul_add(x.e0, x.fac, y.e0, y.fac);
        if (MODE != 1) r1 = r1 + F::mul_add(x.q, x.fac, y.q, y.fac);
      } else if (base + l0 < h) {
        const EqTerm<FID, MODE> x = eq_term<FID, MODE>(A, B, C, nullptr, eqR, 0, 0, h, nk, base + l0, l0);
        r0 = r0 + x.e0 * x.fac;
        if (MODE != 1) r1 = r1 + x.q * x.fac;
      }
      if (++pend == 6) {
        r0 = r0.norm().canon();
        r1 = r1.norm().canon();
        pend = 0;
      }
    }
    if (k < K) {
      const uint32_t l0 = lo0 + k * P;
      if (base + l0 < h) {
        const EqTerm<FID, MODE> x = eq_term<FID, MODE>(A, B, C, nullptr, eqR, 0, 0, h, nk, base + l0, l0);
        r0 = r0 + x.e0 * x.fac;
        if (MODE != 1) r1 = r1 + x.q * x.fac;
      }
    }
    const F el = ldw<FID>(eqL, hi);
    g0 = g0 + r0.norm().canon() * el;
    if (MODE != 1) g1 = g1 + r1.norm().canon() * el;
    if (++pend_g == 6) {
      g0 = g0.norm().canon();
      g1 = g1.norm().canon();
      pend_g = 0;
    }
  }
  g0 = g0.norm().canon();
  g1 = g1.norm().canon();
  if constexpr (MODE == 1) {  // one sum only (g1 stays zero): at 2^20 the block sum is 40 % of this kernel's instructions
    F x[1] = {g0};
    block_sum_waves<FID, 1>(x, lds);
    g0 = x[0];
  } else {
    block_sum_pair<FID>(g0, g1, lds);
  }
  if (threadIdx.x == 0) {
    g0.to_words(partial + 16 * blockIdx.x);
    g1.to_words(partial + 16 * blockIdx.x + 8);
  }
}

template <int FID> static Fp<FID> challenge_internal(const void* r, bool mont) {
  uint32_t w[8];
  memcpy(w, r, 32);
  require(Fp<FID>::words_lt_p(w), NMX_E_SCALAR_RANGE, "challenge >= field modulus");
  Fp<FID> f = Fp<FID>::from_words(w);
  return (mont ? f.mont256_to_internal() : f.to_internal()).canon();
}

// 2^e mod p as a plain integer in limbs (host)
template <int FID> static Fp<FID> pow2_plain(uint32_t e) {
  using F = Fp<FID>;
  F two = F::zero();
  two.l[0] = 2;
  F base = two.to_internal().canon(), acc = F::one();
  for (int i = 31; i >= 0; i--) {
    acc = acc.sqr();
    if ((e >> i) & 1u) acc = acc * base;
  }
  return acc.to_canonical();
}

template <int FID, int MODE>
static void eq_sums_t(Ctx& c, const void* A, const void* B, const void* C, size_t len, const void* eqL, size_t nL,
                      const void* eqR, size_t nR, uint32_t shift, uint32_t flags, uint8_t* out) {
  using F = Fp<FID>;
  const bool mont = flags & NMX_SCALARS_MONT, dev = flags & NMX_SCALARS_DEVICE;
  const uint32_t h = (uint32_t)(len / 2);
  // >= 8 indices per lane before the (comparatively expensive) LDS tree, at most 2048 blocks
  const uint32_t want = (h + 256 * 8 - 1) / (256 * 8);
  // grid cap: the chip holds 768 blocks of these kernels at three waves per SIMD; mode 1 (evaluate_with) runs best with exactly
  // that many (2^24: 0.140 ms against 0.150 at 2048, 0.168 at 4096), modes 2 / 3 are flat from 768 to 3072 (0.404-0.413 ms) and
  // keep 2048 (measured on one lease, second half of round 3; NMX_TUNE_EQ_MAX_BLOCKS / option eq_max_blocks)
  const uint32_t cap = G.eq_max_blocks ? (uint32_t)G.eq_max_blocks : (MODE == 1 ? 768u : 2048u);
  const uint32_t blocks = want < 1 ? 1 : (want > cap ? cap : want);
  // staging (host operands) + partials + result
  size_t need = (size_t)blocks * 64 + 64 + 512;
  auto pad = [](size_t b) { return (b + 255) & ~(size_t)255; };
  if (!dev) need += pad(len * 32) * (MODE == 3 ? 3 : MODE == 2 ? 2 : 1) + pad(nL * 32) + pad(nR * 32);
  arena_reserve(c, need);
  size_t used = 0;
  auto stage = [&](const void* p, size_t elems) -> const uint32_t* {
    if (!p) return nullptr;
    if (dev) return (const uint32_t*)p;
    char* d = c.arena + used;
    used += pad(elems * 32);
    HIPCHK(hipMemcpyAsync(d, p, elems * 32, hipMemcpyHostToDevice, c.stream));
    return (const uint32_t*)d;
  };
  const uint32_t* dA = stage(A, len);
  const uint32_t* dB = MODE >= 2 ? stage(B, len) : nullptr;
  const uint32_t* dC = MODE == 3 ? stage(C, len) : nullptr;
  const uint32_t* dL = stage(eqL, nL);
  const uint32_t* dR = stage(eqR, nR);
  uint32_t* partial = (uint32_t*)(c.arena + used);
  used += pad((size_t)blocks * 64);
  uint32_t* dout = (uint32_t*)(c.arena + used);
  // Form factor Fm = 1 (canonical) or 2^256 (Montgomery); R' = 2^261; a product of two stored elements comes out
  // as x*y * Fm^2 / R'.  MODE 3 brings c0*Fm to that scale with the plain constant Fm; MODE 2 subtracts the
  // constant 1 * Fm^2 / R' directly.
  F fconst = F::zero();
  if (MODE == 3) {
    if (mont) fconst = pow2_plain<FID>(256);
    else fconst.l[0] = 1;
  } else if (MODE == 2) {
    F one_plain = F::zero();
    one_plain.l[0] = 1;
    fconst = mont ? pow2_plain<FID>(512 - 261) : one_plain.to_canonical();  // 2^251, or 2^-261 mod p
  }
  // the kernels add nk = p - fconst (the subtrahend's negative, a canonical field element)
  const F nk = MODE == 1 ? F::zero() : F::sub2(F::zero(), fconst.canon()).norm().canon();
  const bool prof = G.profiling;
  DeviceBackend be(c, false, prof);
  be.mark("k");
  const uint32_t mask = shift >= 32 ? 0xffffffffu : ((1u << shift) - 1u);
  if (dL && shift < 31) {
    hipLaunchKernelGGL((k_eq_rows<FID, MODE>), dim3(blocks), dim3(256), 0, c.stream, dA, dB, dC, dL, dR, shift, h, nk, partial);
  } else {
    hipLaunchKernelGGL((k_eq_sums<FID, MODE>), dim3(blocks), dim3(256), 0, c.stream, dA, dB, dC, dL, dR, shift, mask, h,
                       nk, partial);
  }
  HIPCHK(hipGetLastError());
  if (MODE == 1) hipLaunchKernelGGL((k_sum_partials_n<FID, 1, 2>), dim3(1), dim3(256), 0, c.stream, partial, blocks, dout);  // one sum
  else hipLaunchKernelGGL((k_sum_partials_n<FID, 2, 2>), dim3(1), dim3(256), 0, c.stream, partial, blocks, dout);
  be.mark("end");
  uint32_t res[16];
  be.d2h(res, dout, MODE == 1 ? 32 : 64);  // through the context's pinned landing buffer
  be.sync();
  if (MODE == 1) memset(res + 8, 0, 32);
  if (prof && be.nmarks == 2) {
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, c.ev[0], c.ev[1]));
    prof_store(&ms, 1);
  }
  // sums carry x * Fm^k / R'^(k-1) with k = 4 (modes 2, 3: two data factors + two eq factors) or, without eqL, k = 3;
  // mode 1: k = 3 / 2.  Bring them back to the vectors' own form x * Fm:  multiply by R'^(k-1) / Fm^(k-1)
  // (one more Montgomery product => the plain constant is R'^k / Fm^(k-1)).
  const uint32_t k = (MODE == 1 ? 2u : 3u) + (eqL ? 1u : 0u);
  const uint32_t e = 261u * k - (mont ? 256u * (k - 1) : 0u);
  F corr = pow2_plain<FID>(e);
  for (int j = 0; j < 2; j++) {
    F v = F::from_words(res + 8 * j) * corr;
    uint32_t w[8];
    v.canon().to_words(w);
    memcpy(out + 32 * j, w, 32);
  }
}

// ---- one whole prover round in one pass: bind with the challenge AND the next round's eq-factored sums --------------
// The reference binds A, B, C with the round challenge (bind_poly_var_top, multilinear.rs:65-84, via
// sumcheck.rs:535-545) and then, at the top of the next round, streams the bound tables again for the evaluation
// points (sumcheck.rs:900-1075).  The bound values a lane has just computed ARE the next round's operands: with
// hq = len/4, lane id binds X[id], X[id + hq] (from X[id + 2hq], X[id + 3hq]) and these are (x0, x1) of next-round
// index id.  One pass reads every table once and writes the bound halves: 576 B per index for three tables instead
// of 576 B (three binds) + 320 B (sums) -- the sums ride along on the multiplier while the pass waits for HBM.
// Binding in place is safe: a lane reads exactly the two low-half elements it overwrites.
template <int FID, int MODE>
__global__ __launch_bounds__(256) void k_bind_eq_sums(const uint32_t* A, const uint32_t* B, const uint32_t* C,
                                                      uint32_t* oA, uint32_t* oB, uint32_t* oC, Fp<FID> r,
                                                      const uint32_t* eqL, const uint32_t* eqR, uint32_t shift,
                                                      uint32_t mask, uint32_t hq, Fp<FID> nk, uint32_t* partial) {
  using F = Fp<FID>;
  __shared__ uint32_t lds[9 * 256];
  F s0 = F::zero(), s1 = F::zero();
  uint32_t pending = 0;
  auto bind2 = [&](const uint32_t* X, uint32_t* oX, uint32_t id, F& y0, F& y1) {
    const F x00 = ldw<FID>(X, id), x01 = ldw<FID>(X, (size_t)id + hq);
    const F x10 = ldw<FID>(X, (size_t)id + 2 * (size_t)hq), x11 = ldw<FID>(X, (size_t)id + 3 * (size_t)hq);
    y0 = (x00 + r * F::sub2(x10, x00).norm()).norm().canon();   // lo + r * (hi - lo), as BindTopFn
    y1 = (x01 + r * F::sub2(x11, x01).norm()).norm().canon();
    y0.to_words(oX + 8 * (size_t)id);
    y1.to_words(oX + 8 * ((size_t)id + hq));
  };
  for (uint32_t id = blockIdx.x * 256u + threadIdx.x; id < hq; id += gridDim.x * 256u) {
    F a0, a1, b0, b1, c0, c1;
    bind2(A, oA, id, a0, a1);
    if (MODE >= 2) bind2(B, oB, id, b0, b1);
    if (MODE == 3) bind2(C, oC, id, c0, c1);
    F fac = ldw<FID>(eqR, eqL ? (id & mask) : id);
    if (eqL) fac = ldw<FID>(eqL, id >> shift) * fac;
    if (MODE == 1) {
      s0 = s0 + a0 * fac;
    } else {
      const F e0 = MODE == 3 ? F::mul_add(a0, b0, c0, nk) : (a0 * b0 + nk).norm();  // a0*b0 - c0*k, one reduction
      const F q = F::sub2(a1, a0).norm() * F::sub2(b1, b0).norm();
      s0 = s0 + e0 * fac;
      s1 = s1 + q * fac;
    }
    if (++pending == 6) {
      s0 = s0.norm().canon();
      s1 = s1.norm().canon();
      pending = 0;
    }
  }
  s0 = block_sum<FID>(s0.norm().canon(), lds);
  if (MODE != 1) {
    __syncthreads();
    s1 = block_sum<FID>(s1.norm().canon(), lds);
  }
  if (threadIdx.x == 0) {
    s0.to_words(partial + 16 * blockIdx.x);
    s1.to_words(partial + 16 * blockIdx.x + 8);
  }
}


template <int FID, int MODE>
static void bind_eq_sums_t(Ctx& c, const void* A, const void* B, const void* C, size_t len, const void* r, const void* eqL,
                           size_t nL, const void* eqR, size_t nR, uint32_t shift, uint32_t flags, void* oA, void* oB,
                           void* oC, uint8_t* out) {
  using F = Fp<FID>;
  const bool mont = flags & NMX_SCALARS_MONT;
  const uint32_t hq = (uint32_t)(len / 4);
  // four indices per lane at 2^24 (the LDS tree per block is then a small share); below 2^20 indices ONE per lane: a pass is a
  // latency chain of ~10 products per index and four of them in a row made every size from 2^10 to 2^18 take 45-59 us
  const uint32_t per = hq <= (1u << 20) ? 1u : 4u;
  const uint32_t want = (hq + 256 * per - 1) / (256 * per);
  const uint32_t blocks = want < 1 ? 1 : (want > 4096 ? 4096 : want);
  auto pad = [](size_t b) { return (b + 255) & ~(size_t)255; };
  arena_reserve(c, pad((size_t)blocks * 64) + 64 + 512);
  uint32_t* partial = (uint32_t*)c.arena;
  uint32_t* dout = (uint32_t*)(c.arena + pad((size_t)blocks * 64));
  F fconst = F::zero();
  if (MODE == 3) {
    if (mont) fconst = pow2_plain<FID>(256);
    else fconst.l[0] = 1;
  } else if (MODE == 2) {
    F one_plain = F::zero();
    one_plain.l[0] = 1;
    fconst = mont ? pow2_plain<FID>(512 - 261) : one_plain.to_canonical();
  }
  const F nk = MODE == 1 ? F::zero() : F::sub2(F::zero(), fconst.canon()).norm().canon();  // p - fconst, as eq_sums_t
  const F ri = challenge_internal<FID>(r, mont);
  const bool prof = G.profiling;
  DeviceBackend be(c, false, prof);
  be.mark("k");
  const uint32_t mask = shift >= 32 ? 0xffffffffu : ((1u << shift) - 1u);
  hipLaunchKernelGGL((k_bind_eq_sums<FID, MODE>), dim3(blocks), dim3(256), 0, c.stream, (const uint32_t*)A,
                     (const uint32_t*)B, (const uint32_t*)C, (uint32_t*)oA, (uint32_t*)oB, (uint32_t*)oC, ri,
                     (const uint32_t*)eqL, (const uint32_t*)eqR, shift, mask, hq, nk, partial);
  HIPCHK(hipGetLastError());
  hipLaunchKernelGGL((k_sum_partials_n<FID, 2, 2>), dim3(1), dim3(256), 0, c.stream, partial, blocks, dout);
  be.mark("end");
  uint32_t res[16];
  be.d2h(res, dout, 64);  // through the context's pinned landing buffer
  be.sync();
  if (prof && be.nmarks == 2) {
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, c.ev[0], c.ev[1]));
    prof_store(&ms, 1);
  }
  const uint32_t k = (MODE == 1 ? 2u : 3u) + (eqL ? 1u : 0u);  // as eq_sums_t
  const uint32_t e = 261u * k - (mont ? 256u * (k - 1) : 0u);
  F corr = pow2_plain<FID>(e);
  for (int j = 0; j < 2; j++) {
    F v = F::from_words(res + 8 * j) * corr;
    uint32_t w[8];
    v.canon().to_words(w);
    memcpy(out + 32 * j, w, 32);
  }
  (void)nL;
  (void)nR;
}

void fv_bind_eq_sums(Ctx& c, int field, int mode, const void* A, const void* B, const void* C, size_t len, const void* r,
                     const void* eqL, size_t nL, const void* eqR, size_t nR, uint32_t shift, uint32_t flags, void* oA,
                     void* oB, void* oC, uint8_t* out) {
#define BES(FID)                                                                                                    \
  switch (mode) {                                                                                                   \
    case 1: bind_eq_sums_t<FID, 1>(c, A, B, C, len, r, eqL, nL, eqR, nR, shift, flags, oA, oB, oC, out); return;    \
    case 2: bind_eq_sums_t<FID, 2>(c, A, B, C, len, r, eqL, nL, eqR, nR, shift, flags, oA, oB, oC, out); return;    \
    case 3: bind_eq_sums_t<FID, 3>(c, A, B, C, len, r, eqL, nL, eqR, nR, shift, flags, oA, oB, oC, out); return;    \
    default: throw Fail{NMX_E_ARG, "bad sum-check mode"};                                                           \
  }
  switch (field) {
    case 0: BES(0)
    case 1: BES(1)
    case 2: BES(2)
    case 3: BES(3)
    default: throw Fail{NMX_E_ARG, "bad field id"};
  }
#undef BES
}

// ---- k polynomials evaluated at m points in one launch ------------------------------------------------------------
// HyperKZG's evaluation matrix v[j][i] = f_i(u_j) (/root/reference/src/provider/hyperkzg.rs:1011-1020, 1049-1056):
// ell polynomials of lengths n, n/2, ..., 2 at the three points r, -r, r^2.  One Horner pass per (polynomial, point)
// is 3*ell launches each bound by a 64-step dependent chain; here a lane evaluates a 16-coefficient chunk at all m
// points (coefficients read once), scales by u^(16 * lane) from a table, the block sums over its 256 chunks, and
// thread 0 applies the block's own power u^(4096 * block).  A polynomial owns whole blocks.
static constexpr uint32_t kEvalChunk = 16, kEvalMaxPts = 4;
struct EvalPoly {
  const uint32_t* f;
  uint32_t len, first_block, nblocks;
};
template <int FID>
__global__ __launch_bounds__(256) void k_eval_multi(const EvalPoly* polys, uint32_t k, uint32_t m,
                                                    const uint32_t* pts /* m x 8, internal form */,
                                                    const uint32_t* pw16 /* m x 256 x 8: u^(16 t) */,
                                                    const uint32_t* pw4096 /* m x 20 x 8: u^(4096 * 2^s) */,
                                                    uint32_t* partial /* blocks x kEvalMaxPts x 8 */) {
  using F = Fp<FID>;
  __shared__ uint32_t lds[9 * 256];
  __shared__ uint32_t s_poly;
  if (threadIdx.x == 0) {
    uint32_t p = 0;
    while (p + 1 < k && polys[p + 1].first_block <= blockIdx.x) p++;
    s_poly = p;
  }
  __syncthreads();
  const EvalPoly P = polys[s_poly];
  const uint32_t lb = blockIdx.x - P.first_block;                     // block within the polynomial
  const uint32_t lo = (lb * 256u + threadIdx.x) * kEvalChunk;          // this lane's first coefficient
  F h[kEvalMaxPts];
#pragma unroll
  for (uint32_t j = 0; j < kEvalMaxPts; j++) h[j] = F::zero();
  if (lo < P.len) {
    const uint32_t hi = lo + kEvalChunk < P.len ? lo + kEvalChunk : P.len;
    F u[kEvalMaxPts];
#pragma unroll
    for (uint32_t j = 0; j < kEvalMaxPts; j++) u[j] = j < m ? ldw<FID>(pts, j) : F::zero();
    for (uint32_t i = hi; i-- > lo;) {
      const F c = ldw<FID>(P.f, i);
#pragma unroll
      for (uint32_t j = 0; j < kEvalMaxPts; j++)
        if (j < m) h[j] = (c + u[j] * h[j]).norm();                    // < 2.1 p, normalized
    }
#pragma unroll
    for (uint32_t j = 0; j < kEvalMaxPts; j++)
      if (j < m) h[j] = (h[j] * ldw<FID>(pw16, (size_t)j * 256 + threadIdx.x)).canon();
  }
#pragma unroll  // (constant indices: h[] stays in registers -- as a rolled loop over j < m it lived in 160 B of scratch)
  for (uint32_t j = 0; j < kEvalMaxPts; j++) {
    if (j >= m) break;  // block-uniform
    F sum = block_sum<FID>(h[j], lds);
    if (threadIdx.x == 0) {
      // times u^(4096 * lb): product over the set bits of lb
      for (uint32_t sft = 0; sft < 20; sft++)
        if ((lb >> sft) & 1u) sum = (sum * ldw<FID>(pw4096, (size_t)j * 20 + sft)).canon();
      sum.to_words(partial + ((size_t)blockIdx.x * kEvalMaxPts + j) * 8);
    }
    __syncthreads();
  }
}
// out[p][j] = sum over the polynomial's blocks
template <int FID>
__global__ __launch_bounds__(256) void k_eval_finish(const EvalPoly* polys, uint32_t m, const uint32_t* partial,
                                                     uint32_t* out /* k x m x 8 */) {
  using F = Fp<FID>;
  __shared__ uint32_t lds[9 * 256];
  const EvalPoly P = polys[blockIdx.x];
  for (uint32_t j = 0; j < m; j++) {
    F s = F::zero();
    for (uint32_t b = threadIdx.x; b < P.nblocks; b += 256)
      s = (s + ldw<FID>(partial, ((size_t)(P.first_block + b) * kEvalMaxPts + j))).norm().canon();
    s = block_sum<FID>(s, lds);
    if (threadIdx.x == 0) s.to_words(out + ((size_t)blockIdx.x * m + j) * 8);
    __syncthreads();
  }
}

template <int FID>
static void eval_multi_t(Ctx& c, const void* const* polys, const size_t* lens, size_t k, const void* points, size_t m,
                         uint32_t flags, uint8_t* out) {
  using F = Fp<FID>;
  const bool mont = flags & NMX_SCALARS_MONT, dev = flags & NMX_SCALARS_DEVICE;
  auto pad = [](size_t b) { return (b + 255) & ~(size_t)255; };
  std::vector<EvalPoly> desc(k);
  uint32_t blocks = 0;
  size_t stage_bytes = 0;
  for (size_t i = 0; i < k; i++) {
    const uint32_t nb = (uint32_t)((lens[i] + 256 * kEvalChunk - 1) / (256 * kEvalChunk));
    desc[i] = EvalPoly{nullptr, (uint32_t)lens[i], blocks, nb ? nb : 1};
    blocks += desc[i].nblocks;
    if (!dev) stage_bytes += pad(lens[i] * 32);
  }
  const size_t tab_bytes = pad(m * 32) + pad(m * 256 * 32) + pad(m * 20 * 32);
  const size_t need = stage_bytes + pad(k * sizeof(EvalPoly)) + tab_bytes + pad((size_t)blocks * kEvalMaxPts * 32) +
                      pad(k * m * 32) + 512;
  arena_reserve(c, need);
  size_t used = 0;
  auto take = [&](size_t bytes) {
    char* d = c.arena + used;
    used += pad(bytes);
    return d;
  };
  for (size_t i = 0; i < k; i++) {
    if (dev) {
      desc[i].f = (const uint32_t*)polys[i];
    } else {
      char* d = take(lens[i] * 32);
      if (lens[i]) HIPCHK(hipMemcpyAsync(d, polys[i], lens[i] * 32, hipMemcpyHostToDevice, c.stream));
      desc[i].f = (const uint32_t*)d;
    }
  }
  // tables on the host: ~280 multiplications per point, in 4 x 64-bit Montgomery arithmetic (HostFp4: ~25 ns a product; the
  // portable build of the device's 9 x 29-bit form costs ~250 ns -- 0.2 ms of table building in front of a 30-60 us kernel,
  // which is what a HyperKZG prove's evaluation matrix cost until round 6)
  using H = HostFp4<FID>;
  std::vector<uint32_t> hpts(8 * m), h16(8 * 256 * m), h4096(8 * 20 * m);
  for (size_t j = 0; j < m; j++) {
    const uint8_t* pj = (const uint8_t*)points + 32 * j;
    (void)challenge_internal<FID>(pj, mont);               // (the range check: a point >= p is NMX_E_SCALAR_RANGE)
    const H u = mont ? H::from_mont256(pj) : H::from_canonical(pj);
    u.to_device().to_words(hpts.data() + 8 * j);
    H u16 = u;
    for (int q = 0; q < 4; q++) u16 = u16 * u16;           // u^16
    H pw = H::one();
    for (int t = 0; t < 256; t++) {
      pw.to_device().to_words(h16.data() + 8 * (j * 256 + t));
      pw = pw * u16;
    }
    H big = pw;  // u^(16 * 256) = u^4096
    for (int sft = 0; sft < 20; sft++) {
      big.to_device().to_words(h4096.data() + 8 * (j * 20 + sft));
      big = big * big;
    }
  }
  // descriptors and the three tables travel as ONE host-to-device copy (they sit back to back in the arena; four copies from
  // pageable memory were four staged transfers of 10-20 us each in front of a 30 us kernel)
  const size_t o_desc = 0, o_pts = o_desc + pad(k * sizeof(EvalPoly)), o_16 = o_pts + pad(m * 32), o_4096 = o_16 + pad(m * 256 * 32),
               up_bytes = o_4096 + pad(m * 20 * 32);
  char* d_up = take(up_bytes);
  EvalPoly* d_desc = (EvalPoly*)(d_up + o_desc);
  uint32_t* d_pts = (uint32_t*)(d_up + o_pts);
  uint32_t* d_16 = (uint32_t*)(d_up + o_16);
  uint32_t* d_4096 = (uint32_t*)(d_up + o_4096);
  uint32_t* d_part = (uint32_t*)take((size_t)blocks * kEvalMaxPts * 32);
  uint32_t* d_out = (uint32_t*)take(k * m * 32);
  std::vector<char> up(up_bytes, 0);
  memcpy(up.data() + o_desc, desc.data(), k * sizeof(EvalPoly));
  memcpy(up.data() + o_pts, hpts.data(), m * 32);
  memcpy(up.data() + o_16, h16.data(), m * 256 * 32);
  memcpy(up.data() + o_4096, h4096.data(), m * 20 * 32);
  HIPCHK(hipMemcpyAsync(d_up, up.data(), up_bytes, hipMemcpyHostToDevice, c.stream));
  const bool prof = G.profiling;
  DeviceBackend be(c, false, prof);
  be.mark("k");
  hipLaunchKernelGGL((k_eval_multi<FID>), dim3(blocks), dim3(256), 0, c.stream, d_desc, (uint32_t)k, (uint32_t)m, d_pts,
                     d_16, d_4096, d_part);
  HIPCHK(hipGetLastError());
  hipLaunchKernelGGL((k_eval_finish<FID>), dim3((uint32_t)k), dim3(256), 0, c.stream, d_desc, (uint32_t)m, d_part, d_out);
  HIPCHK(hipGetLastError());
  be.mark("end");
  std::vector<uint32_t> res(8 * k * m);
  be.d2h(res.data(), d_out, k * m * 32);
  be.sync();
  if (prof && be.nmarks == 2) {
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, c.ev[0], c.ev[1]));
    prof_store(&ms, 1);
  }
  // the sums carry f * Fm (the coefficients' own form): nothing to correct -- every product had one internal-form factor
  memcpy(out, res.data(), k * m * 32);
}

void fv_eval_multi(Ctx& c, int field, const void* const* polys, const size_t* lens, size_t k, const void* points, size_t m,
                   uint32_t flags, uint8_t* out) {
  switch (field) {
    case 0: eval_multi_t<0>(c, polys, lens, k, points, m, flags, out); break;
    case 1: eval_multi_t<1>(c, polys, lens, k, points, m, flags, out); break;
    case 2: eval_multi_t<2>(c, polys, lens, k, points, m, flags, out); break;
    case 3: eval_multi_t<3>(c, polys, lens, k, points, m, flags, out); break;
    default: throw Fail{NMX_E_ARG, "bad field id"};
  }
}

// ---- sums without an eq factor (the classic sum-check rounds) ------------------------------------------------------
//   kind 1  quad_prod   (sum a0*b0,        sum dA*dB)                          sumcheck.rs:163-186
//   kind 2  linear      (sum a0-b0,        sum A(-1)-B(-1))                    sumcheck.rs:353-378
//   kind 3  quadratic   (sum a0*b0,        sum A(-1)*B(-1))                    sumcheck.rs:380-405
//   kind 4  cubic       (sum a0*b0*c0,     sum dA*dB*dC,   sum A(-1)B(-1)C(-1)) sumcheck.rs:407-443
// with dX = x1 - x0 and X(-1) = 2*x0 - x1.  64 B (two vectors) or 96 B (three) per index against 2-7 modmuls: HBM-bound.
template <int FID, int KIND>
__global__ __launch_bounds__(256) void k_plain_sums(const uint32_t* A, const uint32_t* B, const uint32_t* C, uint32_t h,
                                                    uint32_t* partial /* 32 words per block */) {
  using F = Fp<FID>;
  __shared__ uint32_t lds[9 * 256];
  F s0 = F::zero(), s1 = F::zero(), s2 = F::zero();
  uint32_t pending = 0;
  const uint32_t stride = gridDim.x * 256u;
  uint32_t id = blockIdx.x * 256u + threadIdx.x;
  if constexpr (KIND == 1 || KIND == 3) {
    // two indices per reduction (mul_add: 2 x 81 + 81 multiply-adds instead of 2 x 162)
    for (; (uint64_t)id + stride < h; id += 2 * stride) {
      const uint32_t jd = id + stride;
      const F a0 = ldw<FID>(A, id), a1 = ldw<FID>(A, (size_t)id + h), b0 = ldw<FID>(B, id), b1 = ldw<FID>(B, (size_t)id + h);
      const F c0 = ldw<FID>(A, jd), c1 = ldw<FID>(A, (size_t)jd + h), d0 = ldw<FID>(B, jd), d1 = ldw<FID>(B, (size_t)jd + h);
      s0 = s0 + F::mul_add(a0, b0, c0, d0);
      if (KIND == 1)
        s1 = s1 + F::mul_add(F::sub2(a1, a0).norm(), F::sub2(b1, b0).norm(), F::sub2(c1, c0).norm(), F::sub2(d1, d0).norm());  // operands < 3 p -> < 1.15 p
      else
        s1 = s1 + F::mul_add(F::sub2(a0.dbl(), a1).norm(), F::sub2(b0.dbl(), b1).norm(), F::sub2(c0.dbl(), c1).norm(),
                             F::sub2(d0.dbl(), d1).norm());  // operands < 4 p -> < 1.26 p
      if (++pending == 6) {
        s0 = s0.norm().canon();
        s1 = s1.norm().canon();
        pending = 0;
      }
    }
  }
  for (; id < h; id += stride) {
    const F a0 = ldw<FID>(A, id), a1 = ldw<FID>(A, (size_t)id + h);
    const F b0 = ldw<FID>(B, id), b1 = ldw<FID>(B, (size_t)id + h);
    if (KIND == 1) {
      s0 = s0 + a0 * b0;
      s1 = s1 + F::sub2(a1, a0).norm() * F::sub2(b1, b0).norm();          // operands < 3 p -> < 1.08 p
    } else if (KIND == 2) {
      s0 = s0 + F::sub2(a0, b0);                                          // < 3 p
      s1 = s1 + F::sub4(F::sub2(a0.dbl(), a1).norm(), F::sub2(b0.dbl(), b1).norm());  // < 4p + 4p
    } else if (KIND == 3) {
      s0 = s0 + a0 * b0;
      s1 = s1 + F::sub2(a0.dbl(), a1).norm() * F::sub2(b0.dbl(), b1).norm();  // operands < 4 p -> < 1.13 p
    } else {
      const F c0 = ldw<FID>(C, id), c1 = ldw<FID>(C, (size_t)id + h);
      s0 = s0 + (a0 * b0) * c0;
      s1 = s1 + (F::sub2(a1, a0).norm() * F::sub2(b1, b0).norm()) * F::sub2(c1, c0).norm();
      s2 = s2 + (F::sub2(a0.dbl(), a1).norm() * F::sub2(b0.dbl(), b1).norm()) * F::sub2(c0.dbl(), c1).norm();
    }
    // kind 2 adds up to 8 p per term (limbs < 2^31): canonicalise every term; products add < 1.2 p (limbs < 2^29)
    if (KIND == 2 || ++pending == 6) {
      s0 = s0.norm().canon();
      s1 = s1.norm().canon();
      if (KIND == 4) s2 = s2.norm().canon();
      pending = 0;
    }
  }
  s0 = block_sum<FID>(s0.norm().canon(), lds);
  __syncthreads();
  s1 = block_sum<FID>(s1.norm().canon(), lds);
  if (KIND == 4) {
    __syncthreads();
    s2 = block_sum<FID>(s2.norm().canon(), lds);
  }
  if (threadIdx.x == 0) {
    s0.to_words(partial + 32 * blockIdx.x);
    s1.to_words(partial + 32 * blockIdx.x + 8);
    s2.to_words(partial + 32 * blockIdx.x + 16);
  }
}

template <int FID, int KIND>
static void plain_sums_t(Ctx& c, const void* A, const void* B, const void* C, size_t len, uint32_t flags, uint8_t* out) {
  using F = Fp<FID>;
  const bool mont = flags & NMX_SCALARS_MONT, dev = flags & NMX_SCALARS_DEVICE;
  const uint32_t h = (uint32_t)(len / 2);
  const uint32_t want = (h + 256 * 8 - 1) / (256 * 8);
  const uint32_t blocks = want < 1 ? 1 : (want > 2048 ? 2048 : want);
  auto pad = [](size_t b) { return (b + 255) & ~(size_t)255; };
  size_t need = (size_t)blocks * 128 + 128 + 512;
  if (!dev) need += pad(len * 32) * (KIND == 4 ? 3 : 2);
  arena_reserve(c, need);
  size_t used = 0;
  auto stage = [&](const void* p, size_t elems) -> const uint32_t* {
    if (dev) return (const uint32_t*)p;
    char* d = c.arena + used;
    used += pad(elems * 32);
    HIPCHK(hipMemcpyAsync(d, p, elems * 32, hipMemcpyHostToDevice, c.stream));
    return (const uint32_t*)d;
  };
  const uint32_t* dA = stage(A, len);
  const uint32_t* dB = stage(B, len);
  const uint32_t* dC = KIND == 4 ? stage(C, len) : nullptr;
  uint32_t* partial = (uint32_t*)(c.arena + used);
  used += pad((size_t)blocks * 128);
  uint32_t* dout = (uint32_t*)(c.arena + used);
  const bool prof = G.profiling;
  DeviceBackend be(c, false, prof);
  be.mark("k");
  hipLaunchKernelGGL((k_plain_sums<FID, KIND>), dim3(blocks), dim3(256), 0, c.stream, dA, dB, dC, h, partial);
  HIPCHK(hipGetLastError());
  hipLaunchKernelGGL((k_sum_partials_n<FID, 3, 4>), dim3(1), dim3(256), 0, c.stream, partial, blocks, dout);
  be.mark("end");
  uint32_t res[24];
  be.d2h(res, dout, 96);
  be.sync();
  if (prof && be.nmarks == 2) {
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, c.ev[0], c.ev[1]));
    prof_store(&ms, 1);
  }
  // a product of k stored elements (k - 1 Montgomery products) is x * Fm^k / R'^(k-1); see eq_sums_t
  const uint32_t k = KIND == 2 ? 1u : KIND == 4 ? 3u : 2u;
  F corr = pow2_plain<FID>(261u * k - (mont ? 256u * (k - 1) : 0u));
  for (int j = 0; j < (KIND == 4 ? 3 : 2); j++) {
    F v = F::from_words(res + 8 * j) * corr;
    uint32_t w[8];
    v.canon().to_words(w);
    memcpy(out + 32 * j, w, 32);
  }
}

void fv_plain_sums(Ctx& c, int field, int kind, const void* A, const void* B, const void* C, size_t len, uint32_t flags,
                   uint8_t* out) {
#define PLS(FID)                                                                  \
  switch (kind) {                                                                 \
    case 1: plain_sums_t<FID, 1>(c, A, B, C, len, flags, out); return;            \
    case 2: plain_sums_t<FID, 2>(c, A, B, C, len, flags, out); return;            \
    case 3: plain_sums_t<FID, 3>(c, A, B, C, len, flags, out); return;            \
    case 4: plain_sums_t<FID, 4>(c, A, B, C, len, flags, out); return;            \
    default: throw Fail{NMX_E_ARG, "bad sum-check kind"};                         \
  }
  switch (field) {
    case 0: PLS(0)
    case 1: PLS(1)
    case 2: PLS(2)
    case 3: PLS(3)
    default: throw Fail{NMX_E_ARG, "bad field id"};
  }
#undef PLS
}

void fv_eq_sums(Ctx& c, int field, int mode, const void* A, const void* B, const void* C, size_t len, const void* eqL,
                size_t nL, const void* eqR, size_t nR, uint32_t shift, uint32_t flags, uint8_t* out) {
#define EQS(FID)                                                                                        \
  switch (mode) {                                                                                       \
    case 1: eq_sums_t<FID, 1>(c, A, B, C, len, eqL, nL, eqR, nR, shift, flags, out); return;            \
    case 2: eq_sums_t<FID, 2>(c, A, B, C, len, eqL, nL, eqR, nR, shift, flags, out); return;            \
    case 3: eq_sums_t<FID, 3>(c, A, B, C, len, eqL, nL, eqR, nR, shift, flags, out); return;            \
    default: throw Fail{NMX_E_ARG, "bad sum-check mode"};                                               \
  }
  switch (field) {
    case 0: EQS(0)
    case 1: EQS(1)
    case 2: EQS(2)
    case 3: EQS(3)
    default: throw Fail{NMX_E_ARG, "bad field id"};
  }
#undef EQS
}

}  // namespace nmx

#include "sumcheck_prove.hpp"
#include "ipa.hpp"
