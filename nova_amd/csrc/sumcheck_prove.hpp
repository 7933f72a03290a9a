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
//   * the tables are bound IN PLACE and the bind of round j is fused with the sums of round j + 1: every table is read once per
//     round, ONE launch per round (k_sc_pass: one index per thread, the block that draws the last ticket adds the partials up);
//   * the sums land in a MAILBOX -- a few words of coherent pinned host memory written with system scope, sequence word last --
//     which the host polls: no copy engine, no stream synchronisation in a round;
//   * the O(1) algebra of a round (derive_from_claim_deg2/1, UniPoly::from_evals_deg3/2, evaluate, EqSumCheckInstance::bound)
//     runs on the host in 4 x 64-bit Montgomery arithmetic (host_fp4.hpp, sc_host.hpp: ~1 us per round); the only thing that
//     leaves the library is the transcript step: a callback receives the round polynomial's coefficients and returns the
//     challenge (`transcript.absorb(b"p", &poly); transcript.squeeze(b"c")`, sumcheck.rs:224-227,481-484,315-318);
//   * once the tables hold <= 128 elements (option sc_host_tail) the last device bind lands them in pinned memory and the
//     remaining rounds run on the host (sc_host.hpp sc_tail_rounds): ~300 products against a 20-25 us kernel round trip.
// All sqrt-size eq tables of an instance (poly_eq_left[k], poly_eq_right[k], sumcheck.rs:608-641) are built by one launch into a
// heap layout (table k at offset 2^k) in the context's aux arena.
// Measured at num_cons = 2^20 (profiles/r05_spartan): the three provers 1.08 / 2.15 / 1.02 ms as first written (9 x 29-bit host
// products, four indices per thread, pass + final sum) -> 0.77 / 0.91 / 0.74 ms with the host arithmetic, the tail and one
// index per thread.
#pragma once
#include <array>
#include <thread>

#include <immintrin.h>

#include "sc_host.hpp"

namespace nmx {

// ---- mailbox ----------------------------------------------------------------------------------------------------------
static constexpr uint32_t kMailSlots = 16, kMailSlotWords = 64;  // slot = 256 bytes: word 0 sequence, words 8..31 three field elements
static constexpr uint32_t kScSmallHq = 256;                       // bound halves up to this many indices run as one block (one index per thread)

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
  block_sum_waves<FID, J, true>(s, lds);
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
  block_sum_pair<FID, true>(s0, s1, lds);
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
  // a pass enqueued BEFORE its challenge exists (ScPass::prelaunch): r is not in the arguments -- the kernel waits for the host to
  // write it to this 64-byte line (uncached device memory written through the BAR): four 16-byte pieces, each led by chal_seq --
  // {seq, l0, l1, l2} {seq, l3, l4, l5} {seq, l6, l7, l8} {seq, give-up flag, sum0, sum1} -- with a 64-bit checksum of
  // (seq, flag, l0..l8) in the last two words (chal_sum): a line is accepted when all four pieces show the sequence AND the checksum
  // of what was read matches, so no store granularity is assumed (a write-combined 16-byte store may reach the device in 8-byte halves)
  const uint32_t* chal = nullptr;
  uint32_t chal_seq = 0;
};
// 64-bit checksum of a challenge line (two independent 32-bit multiply-xor chains over the eleven payload words).  Not a MAC: it
// separates "the line the host wrote" from "a mixture of that line and the one before it", whatever the mixture's granularity.
NMX_HD void chal_sum(const uint32_t* l9, uint32_t flag, uint32_t seq, uint32_t& s0, uint32_t& s1) {
  uint32_t a = seq ^ 0x9e3779b9u, b = (seq * 0x85ebca6bu) ^ 0xc2b2ae35u;
  for (int i = 0; i <= 9; i++) {
    const uint32_t w = i < 9 ? l9[i] : flag;
    a = (a ^ w) * 0x01000193u;
    a = (a << 13) | (a >> 19);
    b = (b + w + (uint32_t)i) * 0x27d4eb2fu;
    b ^= b >> 15;
  }
  a ^= a >> 16, a *= 0x7feb352du, a ^= a >> 15;
  b ^= b >> 13, b *= 0x846ca68bu, b ^= b >> 16;
  s0 = a, s1 = b ^ 0x5bd1e995u;  // (an all-zero line never passes: seq is never 0, and sum1 of zeros is not 0 either way)
}
// The challenge of a pre-launched pass.  Thread 0 of every block reads the WHOLE line per poll (four 16-byte loads in flight together;
// ten dependent word reads of pinned host memory made the first version of this 25 us slower than launching late).  The host stores
// the line as four 16-byte pieces that each begin with the sequence word; the pass goes ahead when all four show it AND the line's
// checksum (chal_sum, last two words) matches what was read -- no ordering between the pieces and no store or load granularity is
// assumed: a torn line (a piece half old, half new -- write-combining buffers may be evicted in 8-byte chunks) fails the checksum
// and is simply polled again until the whole line has landed (before round 6 a torn piece would have bound the
// tables IN PLACE with a wrong r and the call would have returned NMX_OK with a proof that does not verify).  A rejected line bumps
// the word at line + 16 (a counter the host reads back for NMX_STAT_SC_TORN_REJECTS; block 0 only, once per wait).  false: the
// host said stop, or nothing came for kChalTimeoutTicks of the 100 MHz wall clock -- the block leaves without touching the tables
// or the mailbox (the host then fails its own wait).
static constexpr uint64_t kChalTimeoutTicks = 200000000ull;  // 2 s
static constexpr uint32_t kChalLineWords = 32;               // a slot's second challenge line (128 bytes on; each line: 16 words + its reject counter)
template <int FID> __device__ __forceinline__ bool sc_challenge(const ScSmallArgs<FID>& a, Fp<FID>& r, uint32_t* s_r /* LDS, 10 words */) {
  r = a.r;
  if (!a.chal) return true;
  if (threadIdx.x == 0) {
    const uint64_t t0 = wall_clock64();
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const volatile u32x4* line = reinterpret_cast<const volatile u32x4*>(a.chal);  // (volatile: re-read every time round)
    uint32_t ok = 0;
    bool rejected = false;
    u32x4 q0, q1, q2, q3;
    for (uint32_t spin = 0;; spin++) {
      q0 = line[0], q1 = line[1], q2 = line[2], q3 = line[3];
      if (q0.x == a.chal_seq && q1.x == a.chal_seq && q2.x == a.chal_seq && q3.x == a.chal_seq) {  // every 16-byte piece shows the new sequence
        const uint32_t l9[9] = {q0.y, q0.z, q0.w, q1.y, q1.z, q1.w, q2.y, q2.z, q2.w};
        uint32_t c0, c1;
        chal_sum(l9, q3.y, a.chal_seq, c0, c1);
        if (c0 == q3.z && c1 == q3.w) {  // ... and the payload is the one the host summed: the line is whole
          ok = q3.y == 0 ? 1u : 0u;
          break;
        }
        if (!rejected && blockIdx.x == 0) {  // a torn line: counted once, polled again
          rejected = true;
          __hip_atomic_fetch_add(const_cast<uint32_t*>(a.chal) + 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
      }
      if ((spin & 63u) == 63u && wall_clock64() - t0 > kChalTimeoutTicks) break;
    }
    s_r[0] = q0.y, s_r[1] = q0.z, s_r[2] = q0.w, s_r[3] = q1.y, s_r[4] = q1.z, s_r[5] = q1.w, s_r[6] = q2.y, s_r[7] = q2.z, s_r[8] = q2.w;
    s_r[9] = ok;
  }
  __syncthreads();
  if (!s_r[9]) return false;
#pragma unroll
  for (int i = 0; i < 9; i++) r.l[i] = s_r[i];
  return true;
}
// ---- FOUR lanes per index (QUAD): the short chain --------------------------------------------------------------------------
// A small pass is one index per thread and pure latency: 6 binds + 5 products in a row for the cubic prover (~1.2 us each at one
// wave per SIMD), 4 + 2 for quad_prod.  With four lanes per index the binds run side by side (lane t binds table t; the fourth lane of
// the cubic form builds the eq factor), the bound values cross the quad in shuffles, lane 0 forms the t(0) term and lane 1 the t(inf)
// term: 2 + 1.5 + 1 products deep instead of 11 (cubic), 1 + 1 instead of 6 (quad_prod).  Every lane runs the same instructions
// (operands are SELECTED per role, never branched on), so the quad costs one lane's time.  Only for passes of <= kScQuadMaxHq
// indices: beyond that a pass leaves the latency regime and four lanes per index would be four times the work.
static constexpr uint32_t kScQuadMaxHq = 1u << 12;  // (64 blocks of 64 indices: the largest pass whose partial sums the host adds up)
template <int FID> __device__ __forceinline__ Fp<FID> sc_sel(bool c, const Fp<FID>& x, const Fp<FID>& y) {
  Fp<FID> r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.l[i] = c ? x.l[i] : y.l[i];
  return r;
}
template <int FID> __device__ __forceinline__ Fp<FID> sc_quad_get(const Fp<FID>& x, int src) {
  Fp<FID> r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.l[i] = (uint32_t)__shfl((int)x.l[i], src, 4);
  return r;
}
// one index of the cubic prover (MODE 3) on a quad; returns this lane's term (lane 0: the t(0) term, lane 1: the t(inf) term, others: zero)
template <int FID> __device__ __forceinline__ Fp<FID> sc_index_quad3(const ScSmallArgs<FID>& a, uint32_t id, uint32_t role) {
  using F = Fp<FID>;
  const uint32_t* T = role == 0 ? a.A : role == 1 ? a.B : a.C;
  uint32_t* oT = role == 0 ? a.oA : role == 1 ? a.oB : a.oC;
  const bool tab = role < 3;
  F y0, y1;
  if (a.bind) {
    // lanes 0-2: y = lo + r (hi - lo) twice; lane 3: eqL * eqR (ONE * eqR without eqL) and a product nobody reads
    F w0 = F::zero(), w1 = F::zero(), v0 = F::zero(), v1 = F::zero(), u0 = a.r;
    if (tab) {
      const F x00 = ldw<FID>(T, id), x01 = ldw<FID>(T, (size_t)id + a.hq);
      const F x10 = ldw<FID>(T, (size_t)id + 2 * (size_t)a.hq), x11 = ldw<FID>(T, (size_t)id + 3 * (size_t)a.hq);
      w0 = x00, w1 = x01, v0 = F::sub2(x10, x00).norm(), v1 = F::sub2(x11, x01).norm();
    } else {
      v0 = ldw<FID>(a.eqR, a.eqL ? (id & a.mask) : id);
      u0 = a.eqL ? ldw<FID>(a.eqL, id >> a.shift) : F::one();
    }
    const F m0 = u0 * v0, m1 = a.r * v1;
    y0 = (w0 + m0).norm().canon();
    y1 = (w1 + m1).norm().canon();
    if (tab) {
      y0.to_words(oT + 8 * (size_t)id);
      y1.to_words(oT + 8 * ((size_t)id + a.hq));
    }
  } else {
    if (tab) {
      y0 = ldw<FID>(T, id), y1 = ldw<FID>(T, (size_t)id + a.hq);
    } else {
      y0 = ldw<FID>(a.eqR, a.eqL ? (id & a.mask) : id);
      if (a.eqL) y0 = ldw<FID>(a.eqL, id >> a.shift) * y0;  // (one extra product on the no-bind path: the first round of a small instance only)
      y1 = F::zero();
    }
  }
  const F a0 = sc_quad_get<FID>(y0, 0), a1 = sc_quad_get<FID>(y1, 0), b0 = sc_quad_get<FID>(y0, 1), b1 = sc_quad_get<FID>(y1, 1);
  const F c0 = sc_quad_get<FID>(y0, 2), fac = sc_quad_get<FID>(y0, 3);
  // lane 0: a0 b0 - c0 k; lane 1: (a1 - a0)(b1 - b0) -- one instruction stream: the second product of lane 1 is 0 * nk
  const bool first = role == 0;
  const F x1 = sc_sel<FID>(first, a0, F::sub2(a1, a0).norm()), z1 = sc_sel<FID>(first, b0, F::sub2(b1, b0).norm());
  const F x2 = sc_sel<FID>(first, c0, F::zero());
  const F e = F::mul_add(x1, z1, x2, a.nk) * fac;
  return sc_sel<FID>(role < 2, e, F::zero());
}
// one index of quad_prod (MODE 4) on a quad: lane t binds element (t & 1) of table (t >> 1)
template <int FID> __device__ __forceinline__ Fp<FID> sc_index_quad4(const ScSmallArgs<FID>& a, uint32_t id, uint32_t role) {
  using F = Fp<FID>;
  const uint32_t* T = role < 2 ? a.A : a.B;
  uint32_t* oT = role < 2 ? a.oA : a.oB;
  const size_t at = (size_t)id + ((role & 1u) ? a.hq : 0u);
  F y;
  if (a.bind) {
    const F lo = ldw<FID>(T, at), hi = ldw<FID>(T, at + 2 * (size_t)a.hq);
    y = (lo + a.r * F::sub2(hi, lo).norm()).norm().canon();
    y.to_words(oT + 8 * at);
  } else {
    y = ldw<FID>(T, at);
  }
  const F a0 = sc_quad_get<FID>(y, 0), a1 = sc_quad_get<FID>(y, 1), b0 = sc_quad_get<FID>(y, 2), b1 = sc_quad_get<FID>(y, 3);
  const bool first = role == 0;
  const F e = sc_sel<FID>(first, a0, F::sub2(a1, a0).norm()) * sc_sel<FID>(first, b0, F::sub2(b1, b0).norm());
  return sc_sel<FID>(role < 2, e, F::zero());
}
// the quad form of a pass's index loop: quads of the grid stride over the indices; on return s0 / s1 hold this lane's share
template <int FID, int MODE>
__device__ __forceinline__ void sc_quad_loop(const ScSmallArgs<FID>& a, uint32_t first_quad, uint32_t quads, Fp<FID>& s0, Fp<FID>& s1) {
  using F = Fp<FID>;
  static_assert(MODE == 3 || MODE == 4, "quad form: the cubic and the quad_prod provers");
  const uint32_t role = threadIdx.x & 3u;
  F acc = F::zero();
  uint32_t pending = 0;
  for (uint32_t id = first_quad; id < a.hq; id += quads) {  // (the four lanes of a quad share id: the shuffles inside always see all four)
    acc = acc + (MODE == 3 ? sc_index_quad3<FID>(a, id, role) : sc_index_quad4<FID>(a, id, role));
    if (++pending == 6) {
      acc = acc.norm().canon();
      pending = 0;
    }
  }
  acc = acc.norm().canon();
  s0 = sc_sel<FID>(role == 0, acc, F::zero());
  s1 = sc_sel<FID>(role == 1, acc, F::zero());
}

template <int FID, int MODE, bool QUAD = false> __global__ __launch_bounds__(256) void k_sc_small(ScSmallArgs<FID> a_in) {
  ScSmallArgs<FID> a = a_in;  // (a.r is filled in from the challenge line when the pass was pre-launched)
  using F = Fp<FID>;
  __shared__ uint32_t lds[72];
  __shared__ uint32_t s_r[10];
  F s0 = F::zero(), s1 = F::zero();
  if (!sc_challenge<FID>(a_in, a.r, s_r)) return;
  uint32_t pending = 0;
  if constexpr (QUAD) sc_quad_loop<FID, MODE>(a, threadIdx.x >> 2, 64u, s0, s1);
  for (uint32_t id = threadIdx.x; !QUAD && id < a.hq; id += 256u) {
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
  block_sum_pair<FID, true>(s0, s1, lds);
  if (threadIdx.x == 0) {
    s0.to_words(a.slot + 8);
    s1.to_words(a.slot + 16);
    mail_publish(a.slot, a.seq);
  }
}

// ---- the same pass over ANY number of blocks, the final sum inside it (option sc_fused_sum) ------------------------------------
// The block that draws the last of gridDim.x tickets adds the per-block partials up and writes the mailbox: one launch per round
// instead of pass + final sum -- a round's cost at these sizes is launches (5 us of host time and ~8 us of GPU gap + run time
// for the one-block sum: profiles/r05_spartan), not bytes.  Ordering without a device-wide release in every block (an L2
// write-back per block made round 4's passes 8-20 % slower, sumcheck.hip): a block publishes its 16 partial words with
// agent-scope ATOMIC stores (they go through to memory), waits for them (s_waitcnt vmcnt(0)), then takes its ticket with a
// relaxed agent-scope add; the last block reads the partials with agent-scope atomic loads.  The ticket word returns to zero.
template <int FID> struct ScPassArgs {
  ScSmallArgs<FID> s;
  uint32_t* partial;  // 16 words per block
  uint32_t* ticket;
  // A pass of at most kHostPartBlocks blocks skips the ticket and the second block sum: every block writes its two partial sums
  // and the round's sequence number straight into pinned host memory (32 words per block: sums at 0 and 8, sequence at 16) and
  // the HOST adds them up (ScDev::wait) -- a few dozen 256-bit additions there against ~7 us of ticket + reload + block sum +
  // publish at the end of a kernel that is nothing but its dependent chain (timeline: profiles/r05_spartan/timeline_2p20.txt).
  uint32_t* host_part = nullptr;
};
static constexpr uint32_t kHostPartBlocks = 64, kHostPartWords = 32;
template <int FID, int MODE, bool QUAD = false> __global__ __launch_bounds__(256) void k_sc_pass(ScPassArgs<FID> p) {
  using F = Fp<FID>;
  ScSmallArgs<FID> a = p.s;
  __shared__ uint32_t lds[72];
  __shared__ uint32_t s_last;
  __shared__ uint32_t s_r[10];
  F s0 = F::zero(), s1 = F::zero();
  if (!sc_challenge<FID>(p.s, a.r, s_r)) return;  // (no ticket, no partials: the pass never completes and the host's wait fails)
  uint32_t pending = 0;
  if constexpr (QUAD) sc_quad_loop<FID, MODE>(a, blockIdx.x * 64u + (threadIdx.x >> 2), gridDim.x * 64u, s0, s1);
  for (uint32_t id = blockIdx.x * 256u + threadIdx.x; !QUAD && id < a.hq; id += gridDim.x * 256u) {
    F a0, a1, b0 = F::zero(), b1 = F::zero(), c0 = F::zero(), c1;
    sc_bind2<FID>(a.A, a.oA, a.r, id, a.hq, a0, a1);
    if (MODE >= 3) sc_bind2<FID>(a.B, a.oB, a.r, id, a.hq, b0, b1);
    if (MODE == 3) sc_bind2<FID>(a.C, a.oC, a.r, id, a.hq, c0, c1);
    if (MODE == 4) {
      s0 = s0 + a0 * b0;
      s1 = s1 + F::sub2(a1, a0).norm() * F::sub2(b1, b0).norm();
    } else {
      F fac = ldw<FID>(a.eqR, a.eqL ? (id & a.mask) : id);
      if (a.eqL) fac = ldw<FID>(a.eqL, id >> a.shift) * fac;
      if (MODE == 1) {
        s0 = s0 + a0 * fac;
      } else {
        const F e0 = F::mul_add(a0, b0, c0, a.nk);
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
  block_sum_pair<FID, true>(s0, s1, lds);
  if (p.host_part) {  // kernel-uniform
    if (threadIdx.x == 0) {
      uint32_t* mine = p.host_part + kHostPartWords * (size_t)blockIdx.x;
      s0.to_words(mine), s1.to_words(mine + 8);
      mail_publish(mine + 16, a.seq);
    }
    return;
  }
  if (threadIdx.x == 0) {
    uint32_t w[16];
    s0.to_words(w), s1.to_words(w + 8);
    uint32_t* mine = p.partial + 16 * (size_t)blockIdx.x;
#pragma unroll
    for (int i = 0; i < 16; i++) __hip_atomic_store(mine + i, w[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    s_last = __hip_atomic_fetch_add(p.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1u ? 1u : 0u;
  }
  __syncthreads();
  if (!s_last) return;  // block-uniform
  F t0 = F::zero(), t1 = F::zero();
  pending = 0;
  for (uint32_t i = threadIdx.x; i < gridDim.x; i += 256u) {
    uint32_t w[16];
#pragma unroll
    for (int j = 0; j < 16; j++) w[j] = __hip_atomic_load(p.partial + 16 * (size_t)i + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    t0 = (t0 + F::from_words(w)).norm();
    t1 = (t1 + F::from_words(w + 8)).norm();
    if (++pending == 8) {
      t0 = t0.canon();
      t1 = t1.canon();
      pending = 0;
    }
  }
  t0 = t0.canon();
  t1 = t1.canon();
  __syncthreads();  // lds is reused
  block_sum_pair<FID, true>(t0, t1, lds);
  if (threadIdx.x == 0) {
    __hip_atomic_store(p.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    t0.to_words(a.slot + 8);
    t1.to_words(a.slot + 16);
    mail_publish(a.slot, a.seq);
  }
}

// ---- SEVERAL rounds inside one resident kernel (round 6; option sc_resident) ---------------------------------------------------
// From tables of kResidentMaxLen elements down, a round's pass is a few dozen blocks for a few microseconds; what a round COSTS is
// everything around the pass -- kernel start and end, the queue's dispatch, a fresh poll set-up -- ~25-30 us per round against ~10 us
// of dependent work (profiles/r05_spartan/timeline_2p20.txt: run 15-17 us, gap 9-13 us).  This kernel is launched ONCE (a round
// ahead, like a pre-launched pass) and stays for every remaining device round of the prover:
//     wait for the round's challenge (the checksummed BAR line of sc_challenge)  ->  bind the tables in place  ->  next round's sums
//     ->  per-block partial sums + sequence word into pinned host memory (the host adds them: ScPassArgs::host_part)  ->  wait again
// and ends with the hand-over: the last bind lands the tables in the host's tail area (k_sc_bind_to_host's work).  A block stays as
// long as a pass still needs it (pass i has hq0 >> i indices: 64 per block in the four-lane form, 256 in the one-lane form) and leaves
// after its last one.  The HOST is the grid barrier between two rounds: it sends challenge i + 1 only when every block of pass i has
// published -- each block publishes behind an agent-scope release of ALL its lanes' table stores and a system-scope fence, and
// invalidates its L1 (agent-scope acquire) after every challenge, so the tables another block bound in the round before are read
// from L2.  Cancelling (abort word on the line), the 2 s time-out and the device-wide budget of waiting blocks are sc_challenge's /
// ScDev's; a round that needs the tau = 0 fall-back cancels the kernel first and continues on launched passes.
template <int FID> struct ScResArgs {
  uint32_t *A, *B, *C;
  const uint32_t *heapL, *heapR;                    // eq heaps (ScEqDev layout); MODE 4: unused
  uint32_t first_half, second_half, l, round0;      // pass i is the bind of round round0 + i (its sums are round round0 + i + 1's)
  Fp<FID> nk;
  uint32_t len0, passes, tail;  // tables hold len0 elements at entry; `passes` bind + sums passes; tail: one more bind lands the tables on the host
  uint32_t seq0, cs0;           // pass i publishes with sequence seq0 + i and takes its challenge with sequence cs0 + i
  // TWO challenge lines (kChalLineWords apart), pass i polls line i & 1: the host names the line of pass i + 1 as the one to cancel the
  // moment it has sent challenge i (ScPass::res_send), and a cancellation written there must not overwrite a challenge that some
  // block has not read yet -- with one line, a fall-back in ANOTHER claim of a batch cancelled this claim's kernel by overwriting
  // the challenge it was about to read: the pass never ran and the host's wait for its sums failed
  const uint32_t* chal;
  uint32_t line0;               // the line of pass 0 (0 / 1): the slot's lines alternate across EVERYTHING that waits on them, pre-launched passes included
  uint32_t* host_part;          // kHostPartWords per block
  uint32_t* slot;               // mailbox slot of the hand-over
  uint32_t* host_tab[3];        // tail areas (pinned)
  uint32_t ntab;
};
template <int FID> __device__ __forceinline__ void sc_res_tables(const ScResArgs<FID>& a, uint32_t rnd, ScSmallArgs<FID>& x) {
  if (rnd < a.first_half) {  // ScEqDev::tables
    x.eqL = a.heapL + 8 * ((size_t)1 << (a.first_half - rnd));
    x.eqR = a.heapR + 8 * ((size_t)1 << a.second_half);
    x.shift = a.second_half;
    x.mask = a.second_half >= 32 ? 0xffffffffu : ((1u << a.second_half) - 1u);
  } else {
    x.eqL = nullptr;
    x.eqR = a.heapR + 8 * ((size_t)1 << (a.l - rnd));
    x.shift = 0, x.mask = 0xffffffffu;
  }
}
// the one-lane index loop of a bind + sums pass (k_sc_pass's body)
template <int FID, int MODE>
__device__ __forceinline__ void sc_lane_loop(const ScSmallArgs<FID>& a, uint32_t first, uint32_t stride, Fp<FID>& s0, Fp<FID>& s1) {
  using F = Fp<FID>;
  uint32_t pending = 0;
  for (uint32_t id = first; id < a.hq; id += stride) {
    F a0, a1, b0 = F::zero(), b1 = F::zero(), c0 = F::zero(), c1;
    sc_bind2<FID>(a.A, a.oA, a.r, id, a.hq, a0, a1);
    if (MODE >= 3) sc_bind2<FID>(a.B, a.oB, a.r, id, a.hq, b0, b1);
    if (MODE == 3) sc_bind2<FID>(a.C, a.oC, a.r, id, a.hq, c0, c1);
    if (MODE == 4) {
      s0 = s0 + a0 * b0;
      s1 = s1 + F::sub2(a1, a0).norm() * F::sub2(b1, b0).norm();
    } else {
      F fac = ldw<FID>(a.eqR, a.eqL ? (id & a.mask) : id);
      if (a.eqL) fac = ldw<FID>(a.eqL, id >> a.shift) * fac;
      if (MODE == 1) {
        s0 = s0 + a0 * fac;
      } else {
        const F e0 = F::mul_add(a0, b0, c0, a.nk);
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
}
static constexpr uint32_t kResidentMaxLen = 1u << 14;  // tables of at most this many elements at entry (first pass: 2^12 indices)
template <int FID, int MODE, bool QUAD> __host__ __device__ inline uint32_t sc_res_blocks(uint32_t hq) {
  const uint32_t per = QUAD ? 64u : 256u;
  return hq <= per ? 1u : (hq + per - 1) / per;
}
template <int FID, int MODE, bool QUAD> __global__ __launch_bounds__(256) void k_sc_resident(ScResArgs<FID> p) {
  using F = Fp<FID>;
  __shared__ uint32_t lds[72];
  __shared__ uint32_t s_r[10];
  constexpr uint32_t per = QUAD ? 64u : 256u;
  ScSmallArgs<FID> a;
  a.A = p.A, a.B = p.B, a.C = p.C, a.oA = p.A, a.oB = p.B, a.oC = p.C;
  a.eqL = a.eqR = nullptr, a.shift = 0, a.mask = 0xffffffffu;
  a.r = F::zero(), a.nk = p.nk, a.bind = 1u, a.seq = 0, a.slot = nullptr;
  a.chal = p.chal;
  uint32_t len = p.len0;
  for (uint32_t i = 0; i < p.passes; i++, len >>= 1) {
    a.hq = len >> 2;
    const uint32_t nb = sc_res_blocks<FID, MODE, QUAD>(a.hq);
    if (blockIdx.x >= nb) return;  // (block-uniform; nb only shrinks: this block is not needed again)
    a.chal_seq = p.cs0 + i;
    a.chal = p.chal + (((i + p.line0) & 1u) ? kChalLineWords : 0u);  // two lines, used in turn (see ScResArgs::chal)
    if (MODE != 4) sc_res_tables<FID>(p, p.round0 + i + 1, a);  // the sums of this pass are the NEXT round's: its eq tables
    F r;
    if (!sc_challenge<FID>(a, r, s_r)) return;
    a.r = r;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // the tables other blocks bound in the round before: not from this CU's L1
    F s0 = F::zero(), s1 = F::zero();
    if constexpr (QUAD) sc_quad_loop<FID, MODE>(a, blockIdx.x * per + (threadIdx.x >> 2), nb * per, s0, s1);
    else sc_lane_loop<FID, MODE>(a, blockIdx.x * per + threadIdx.x, nb * per, s0, s1);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");  // every lane's table stores, before the block says "done"
    block_sum_pair<FID, true>(s0, s1, lds);
    if (threadIdx.x == 0) {
      uint32_t* mine = p.host_part + kHostPartWords * (size_t)blockIdx.x;
      s0.to_words(mine), s1.to_words(mine + 8);
      mail_publish(mine + 16, p.seq0 + i);
    }
  }
  if (!p.tail || blockIdx.x != 0) return;
  // the hand-over (k_sc_bind_to_host): tables of `len` elements bound to len / 2, landed in the tail areas
  a.chal_seq = p.cs0 + p.passes;
  a.chal = p.chal + (((p.passes + p.line0) & 1u) ? kChalLineWords : 0u);
  F r;
  if (!sc_challenge<FID>(a, r, s_r)) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  const uint32_t half = len >> 1;
  uint32_t* X[3] = {p.A, p.B, p.C};
  for (uint32_t id = threadIdx.x; id < half; id += 256u) {
    for (uint32_t t = 0; t < p.ntab; t++) {
      const F x0 = ldw<FID>(X[t], id), x1 = ldw<FID>(X[t], (size_t)id + half);
      const F y = (x0 + r * F::sub2(x1, x0).norm()).norm().canon();
      y.to_words(X[t] + 8 * (size_t)id);
      y.to_words(p.host_tab[t] + 8 * (size_t)id);
    }
  }
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) mail_publish(p.slot, p.seq0 + p.passes);
}

// ---- the bind that hands the tables to the host: bound values (or, bind = 0, the tables as they are) into pinned memory ------
// The tail rounds run on the host (sc_host.hpp); with half == 1 this is the LAST bind and the values are the final claims
// (poly_A[0], ..., sumcheck.rs:241-248, 499-506).
template <int FID> struct ScBindOutArgs {
  uint32_t* X[3];
  uint32_t* host[3];  // pinned, device-visible: `half` elements each
  Fp<FID> r;
  uint32_t n, half, bind, seq;
  uint32_t* slot;
};
template <int FID> __global__ __launch_bounds__(256) void k_sc_bind_to_host(ScBindOutArgs<FID> a) {
  using F = Fp<FID>;
  for (uint32_t id = threadIdx.x; id < a.half; id += 256u) {
    for (uint32_t t = 0; t < a.n; t++) {
      const F x0 = ldw<FID>(a.X[t], id);
      F y = x0.canon();
      if (a.bind) {
        const F x1 = ldw<FID>(a.X[t], (size_t)id + a.half);
        y = (x0 + a.r * F::sub2(x1, x0).norm()).norm().canon();
        y.to_words(a.X[t] + 8 * (size_t)id);
      }
      y.to_words(a.host[t] + 8 * (size_t)id);
    }
  }
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) mail_publish(a.slot, a.seq);
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
struct ScProf {  // wall-clock split of one prover call (profiling on): where a round's time goes
  double wait = 0, cb = 0;
  uint32_t launches = 0, rounds = 0, host_rounds = 0;
};
static constexpr uint32_t kTailMaxLog2 = 8, kTailMax = 1u << kTailMaxLog2;  // the longest table the tail takes over
static constexpr size_t kMailBytes = kMailSlots * kMailSlotWords * 4, kTailSlotBytes = (size_t)kTailMax * 32;
static constexpr size_t kPartSlotBytes = (size_t)kHostPartBlocks * kHostPartWords * 4, kChalSlotBytes = 256;

template <int FID> struct ScDev {
  using F = Fp<FID>;
  using H = HostFp4<FID>;
  Ctx& c;
  bool mont;
  ScAlg<FID> alg;
  ScProf prof;
  bool profiling;
  uint32_t tail_len;  // tables of at most this many elements finish on the host (1: only the final values come over)
  ScDev(Ctx& ctx, uint32_t flags) : c(ctx), mont((flags & NMX_SCALARS_MONT) != 0), alg(mont), profiling(G.profiling) {
    const uint32_t t = G.sc_host_tail.load(std::memory_order_relaxed);
    tail_len = 1u << (t > kTailMaxLog2 ? kTailMaxLog2 : t);
    mail_init();
  }
  // raw device sums -> elements: a sum of terms with k stored factors is x Fm^k / R'^(k-1) (eq_sums_t); as an element that is
  // (raw words) x 2^(261 (k - 1) + 256 - 256 k [mont]) / 2^256: ONE Montgomery product with a cached constant
  H raw(const uint32_t* words, uint32_t k) const {
    static const std::array<H, 5> tab[2] = {make_corr(false), make_corr(true)};
    return H::from_plain_times(words, tab[mont ? 1 : 0][k]);
  }
  static std::array<H, 5> make_corr(bool m) {
    std::array<H, 5> t;
    t[0] = H::one();
    for (uint32_t k = 1; k <= 4; k++) t[k] = H::pow2(261u * (k - 1) + 256u - (m ? 256u * k : 0u));
    return t;
  }
  // a stored element (the vectors' own form) -> element
  H stored(const uint32_t* words) const { return mont ? H::from_mont256(words) : H::from_canonical(words); }

  // --- mailbox: [slots: kMailBytes][tail areas: kMailSlots x kTailSlotBytes]
  void mail_init() {
    if (c.mail) return;
    void* p = nullptr;
    const size_t bytes = kMailBytes + kMailSlots * kTailSlotBytes + kMailSlots * kPartSlotBytes;
    HIPCHK(hipHostMalloc(&p, bytes, hipHostMallocCoherent | hipHostMallocMapped));
    memset(p, 0, bytes);
    void* d = nullptr;
    HIPCHK(hipHostGetDevicePointer(&d, p, 0));
    c.mail = (char*)p;
    c.mail_dev = (char*)d;
  }
  uint32_t* slot_dev(uint32_t s) const { return (uint32_t*)c.mail_dev + (size_t)s * kMailSlotWords; }
  uint32_t* tail_dev(uint32_t s) const { return (uint32_t*)(c.mail_dev + kMailBytes + (size_t)s * kTailSlotBytes); }
  const uint32_t* tail_host(uint32_t s) const { return (const uint32_t*)(c.mail + kMailBytes + (size_t)s * kTailSlotBytes); }
  // challenge lines of the pre-launched passes: UNCACHED DEVICE memory that the host writes through the large BAR, so that every
  // block of a waiting pass polls local memory (one poller over PCIe answers in 2.7 us, profiles/r05_spartan/signal_ubench.txt, but
  // the up to 64 blocks of a pass polling pinned host memory together made a pre-launched round 5 us SLOWER than a launched one).
  // nullptr: no large BAR (or the allocation failed) -- then nothing is pre-launched.
  uint32_t* chal_line(uint32_t s) {
    if (!c.chal_tried) {
      c.chal_tried = true;
      int large_bar = 0, cur = 0;  // (the lease made the context's device current on this thread)
      void* p = nullptr;
      if (hipGetDevice(&cur) == hipSuccess && hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, cur) == hipSuccess && large_bar &&
          hipExtMallocWithFlags(&p, kMailSlots * kChalSlotBytes, hipDeviceMallocUncached) == hipSuccess) {
        if (hipMemsetAsync(p, 0, kMailSlots * kChalSlotBytes, c.stream) == hipSuccess && hipStreamSynchronize(c.stream) == hipSuccess) c.chal = (uint32_t*)p;
        else (void)hipFree(p);
      }
      (void)hipGetLastError();
    }
    return c.chal ? c.chal + (size_t)s * (kChalSlotBytes / 4) : nullptr;
  }
  // the CPU's stores to BAR memory are write-combined: a fence between the payload and the sequence word, and one behind it to push it out
  static void chal_store(uint32_t* line, const uint32_t* l, uint32_t abort_word, uint32_t seq, uint32_t s0, uint32_t s1) {
    __m128i* q = reinterpret_cast<__m128i*>(line);  // (256-byte aligned)
    _mm_stream_si128(q + 0, _mm_set_epi32((int)l[2], (int)l[1], (int)l[0], (int)seq));
    _mm_stream_si128(q + 1, _mm_set_epi32((int)l[5], (int)l[4], (int)l[3], (int)seq));
    _mm_stream_si128(q + 2, _mm_set_epi32((int)l[8], (int)l[7], (int)l[6], (int)seq));
    _mm_stream_si128(q + 3, _mm_set_epi32((int)s1, (int)s0, (int)abort_word, (int)seq));
    _mm_sfence();  // push the write-combining buffer out
  }
  static void chal_write(uint32_t* line, const uint32_t* limbs9, uint32_t abort_word, uint32_t seq) {
    static const uint32_t zero9[9] = {0};
    const uint32_t* l = limbs9 ? limbs9 : zero9;
    uint32_t s0, s1;
    chal_sum(l, abort_word, seq, s0, s1);
    if (const uint32_t torn = G.sc_torn_test.load(std::memory_order_relaxed)) {
      // option sc_torn_test (tests only): what a write-combining buffer evicted in 8-byte chunks could leave on the device for a
      // while -- every piece already shows the new sequence word and the new checksum is in place, but the SECOND half of each
      // limb piece is still the old line's (here: the new limbs with bits flipped, so that the stale value is never accidentally
      // right).  The waiting pass must poll past it; the whole line follows `torn` microseconds later.
      uint32_t bad[9];
      for (int i = 0; i < 9; i++) bad[i] = (i % 3) ? (l[i] ^ 0x15a5a5a5u) & 0x1fffffffu : l[i];
      chal_store(line, bad, abort_word, seq, s0, s1);
      const auto t0 = std::chrono::steady_clock::now();
      while (std::chrono::steady_clock::now() - t0 < std::chrono::microseconds(torn)) {
      }
      G.sc_torn_injected.fetch_add(1, std::memory_order_relaxed);
    }
    chal_store(line, l, abort_word, seq, s0, s1);
  }
  // lines the device rejected (the counter word behind each slot's line, bumped by sc_challenge): read back over the BAR -- slow,
  // so only when the torn-line test is on
  void collect_torn_rejects() {
    if (!c.chal || !G.sc_torn_test.load(std::memory_order_relaxed)) return;
    for (uint32_t s_ = 0; s_ < kMailSlots; s_++)
      for (uint32_t line = 0; line < 2; line++) {
        volatile uint32_t* w = c.chal + (size_t)s_ * (kChalSlotBytes / 4) + line * kChalLineWords + 16;
        const uint32_t v = *w;
        if (v) {
          *w = 0;
          _mm_sfence();
          G.sc_torn_rejects.fetch_add(v, std::memory_order_relaxed);
        }
      }
  }
  // Pre-launched passes waiting for a challenge (chal_seq per slot, 0: none).  While one waits, no stream of this call may be
  // synchronised -- the wait would sit behind a kernel that waits for THIS thread -- so the mailbox polls keep polling (up to
  // kArmedPollSeconds, yielding) instead of giving up after sc_poll_us, and anything that must synchronise cancels them first.
  uint32_t armed_seq[kMailSlots] = {};
  uint32_t armed_line[kMailSlots] = {};  // word offset of the line the waiting pass polls (0 or kChalLineWords)
  // A slot's two lines are used in turn by everything that waits on them -- pre-launched passes and the passes of a resident kernel
  // alike: whatever is armed polls the line that the pass before it did NOT use, so a cancellation (written to the armed line) can
  // never overwrite a challenge that an earlier pass has been sent but has not read yet.  next_line[s]: the line the next waiter takes.
  uint32_t next_line[kMailSlots] = {};
  uint32_t take_line(uint32_t s) {
    const uint32_t l = next_line[s];
    next_line[s] ^= 1u;
    return l;
  }
  // Blocks of pre-launched passes sit on CUs doing nothing but polling; forward progress of everything else -- the pass whose
  // sums the host needs before it can send the challenge, other callers' kernels -- needs free slots.  A device-wide budget
  // (kArmedBlocksCap, a quarter of the 1024 blocks of 256 lanes the chip holds at this register count) bounds them: a pass that
  // would exceed it is launched late instead (16 claims x 64 blocks x several concurrent callers).
  uint32_t armed_blocks[kMailSlots] = {};
  static constexpr uint32_t kArmedBlocksCap = 256;
  static std::atomic<uint32_t>& armed_total() {
    static std::atomic<uint32_t> v{0};
    return v;
  }
  bool arm_reserve(uint32_t slot, uint32_t blocks) {
    if (armed_total().fetch_add(blocks, std::memory_order_relaxed) + blocks > kArmedBlocksCap) {
      armed_total().fetch_sub(blocks, std::memory_order_relaxed);
      return false;
    }
    armed_blocks[slot] = blocks;
    return true;
  }
  void arm_release(uint32_t slot) {
    if (armed_blocks[slot]) armed_total().fetch_sub(armed_blocks[slot], std::memory_order_relaxed);
    armed_blocks[slot] = 0;
  }
  static constexpr int kArmedPollSeconds = 4;
  bool any_armed() const {
    for (uint32_t q : armed_seq)
      if (q) return true;
    return false;
  }
  void cancel_armed() noexcept {
    for (uint32_t s = 0; s < kMailSlots; s++)
      if (armed_seq[s]) {
        chal_write(chal_line(s) + armed_line[s], nullptr, 1u, armed_seq[s]);
        armed_seq[s] = 0;
        armed_line[s] = 0;
        arm_release(s);
      }
  }
  ~ScDev() {
    cancel_armed();
    for (uint32_t s = 0; s < kMailSlots; s++) arm_release(s);
  }
  // has a poll that started at t0 gone on long enough?  (spin: the caller's iteration count, to look at the clock only now and then)
  bool poll_over(const std::chrono::steady_clock::time_point& t0, uint32_t spin, uint32_t poll_us) const {
    if ((spin & 1023u) != 1023u) return false;
    const auto el = std::chrono::steady_clock::now() - t0;
    if (!any_armed()) return el > std::chrono::microseconds(poll_us);
    if (el > std::chrono::microseconds(poll_us)) std::this_thread::yield();
    return el > std::chrono::seconds(kArmedPollSeconds);
  }
  // per-block partial sums of the host-summed passes (their own region: a sequence word must never be compared with table data)
  uint32_t* part_dev(uint32_t s) const { return (uint32_t*)(c.mail_dev + kMailBytes + kMailSlots * kTailSlotBytes + (size_t)s * kPartSlotBytes); }
  const uint32_t* part_host(uint32_t s) const {
    return (const uint32_t*)(c.mail + kMailBytes + kMailSlots * kTailSlotBytes + (size_t)s * kPartSlotBytes);
  }
  uint32_t next_seq() { return ++c.mail_seq ? c.mail_seq : ++c.mail_seq; }  // never 0 (a fresh mailbox reads 0)
  // k consecutive sequence numbers, none of them 0 (the resident kernel counts its passes up from a base)
  uint32_t reserve_seq(uint32_t k) {
    if (c.mail_seq > 0xffffffffu - k - 1u) c.mail_seq = 0;
    const uint32_t base = c.mail_seq + 1u;
    c.mail_seq += k;
    return base;
  }
  // waits until slot s carries `seq`; returns its result words.  The sequence word is polled in host memory; the stream is only
  // synchronised when the poll gives up (option sc_poll_us) or polling is off.
  // passes whose per-block partial sums come to the host un-added (ScPassArgs::host_part): blocks expected per slot, and the sums
  uint32_t parts[kMailSlots] = {};
  uint32_t sumbuf[kMailSlots][16];
  const uint32_t* wait_parts(uint32_t s, uint32_t seq, uint32_t nb) {
    const auto t0 = std::chrono::steady_clock::now();
    const uint32_t* area = part_host(s);
    const uint32_t poll_us = G.sc_poll_us.load(std::memory_order_relaxed);
    auto arrived = [&](uint32_t b) { return __atomic_load_n(area + kHostPartWords * (size_t)b + 16, __ATOMIC_ACQUIRE) == seq; };
    bool synced = false;
    H a0 = H::zero(), a1 = H::zero();
    for (uint32_t b = 0; b < nb; b++) {
      if (!synced && !arrived(b)) {
        bool ok = false;
        if (poll_us) {
          for (uint32_t spin = 0;; spin++) {
            if (arrived(b)) {
              ok = true;
              break;
            }
            if (poll_over(t0, spin, poll_us)) break;
          }
        }
        if (!ok) {
          sync_all();
          synced = true;
        }
      }
      if (!arrived(b)) {
        char msg[200];
        snprintf(msg, sizeof msg, "sum-check: a block's partial sums never reached the host (slot %u, block %u of %u, sequence %u, found %u)", s, b, nb, seq,
                 __atomic_load_n(area + kHostPartWords * (size_t)b + 16, __ATOMIC_ACQUIRE));
        throw Fail{NMX_E_HIP, msg};
      }
      a0 = a0 + H::from_mont256(area + kHostPartWords * (size_t)b);  // canonical words: addition does not care about the form
      a1 = a1 + H::from_mont256(area + kHostPartWords * (size_t)b + 8);
    }
    a0.to_mont256(sumbuf[s]), a1.to_mont256(sumbuf[s] + 8);
    if (profiling) prof.wait += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return sumbuf[s];
  }
  const uint32_t* wait(uint32_t s, uint32_t seq) {
    if (const uint32_t nb = parts[s]) {
      parts[s] = 0;
      return wait_parts(s, seq, nb);
    }
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
        if (poll_over(t0, spin, poll_us)) break;
      }
    }
    if (!ok) {
      sync_all();
      require(__atomic_load_n(host, __ATOMIC_ACQUIRE) == seq, NMX_E_HIP, "sum-check: the round's result never reached its mailbox");
    }
    if (profiling) prof.wait += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return host + 8;
  }
  // the context's stream and the side streams the batch prover put work on
  void sync_all() {
    cancel_armed();  // (a waiting pre-launched pass leaves at once; the proof is over if there was one)
    stream_wait(c.stream);
    for (hipStream_t sd : c.side)
      if (sd) HIPCHK(hipStreamSynchronize(sd));
  }
  void sync_all_quiet() noexcept {
    cancel_armed();
    (void)hipStreamSynchronize(c.stream);
    for (hipStream_t sd : c.side)
      if (sd) (void)hipStreamSynchronize(sd);
  }
  // side stream i (created on first use), ordered behind everything on the context's stream so far
  hipStream_t side_stream(uint32_t i) {
    require(i < (uint32_t)Ctx::kSideStreams, NMX_E_HIP, "sum-check: side stream out of range");
    if (!c.side[i]) HIPCHK(hipStreamCreateWithFlags(&c.side[i], hipStreamNonBlocking));
    if (!c.side_ev) HIPCHK(hipEventCreateWithFlags(&c.side_ev, hipEventDisableTiming));
    HIPCHK(hipEventRecord(c.side_ev, c.stream));
    HIPCHK(hipStreamWaitEvent(c.side[i], c.side_ev, 0));
    return c.side[i];
  }
  void launched(uint32_t n = 1) { prof.launches += n; }
  // the transcript step, timed
  H ask(TranscriptFn cb, void* ctx, const H* co, uint32_t n, uint8_t* polys_out, uint8_t* r_out) {
    if (!profiling) return alg.ask(cb, ctx, co, n, polys_out, r_out);
    const auto t0 = std::chrono::steady_clock::now();
    const H r = alg.ask(cb, ctx, co, n, polys_out, r_out);
    prof.cb += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();  // (includes two conversions)
    return r;
  }
  void finish_profile(double total_ms) {
    if (!profiling) return;
    float v[7] = {(float)total_ms, (float)prof.wait, (float)(total_ms - prof.wait - prof.cb), (float)prof.cb, (float)prof.launches,
                  (float)prof.rounds, (float)prof.host_rounds};
    prof_store(v, 7);
  }
};
static void rethrow(const ScFail& f) { throw Fail{f.code == 2 ? NMX_E_SCALAR_RANGE : NMX_E_ARG, "sum-check: " + f.msg}; }

// grids: the first round's sums keep the rules of eq_sums_t / plain_sums_t (tuned at 2^24); the fused bind + sums passes run ONE
// index per thread -- at the sizes a prover sees (<= 2^20 indices) a pass is a latency chain of ~10 products per index, and
// four indices per thread (bind_eq_sums_t's rule, right for 2^24) made every pass from 2^10 to 2^18 indices take 45-59 us
// (profiles/r05_spartan: one block per CU at 2^18)
static inline uint32_t sc_blocks_sums(uint32_t h, bool mode1) {
  const uint32_t want = (h + 256 * 8 - 1) / (256 * 8), cap = G.eq_max_blocks ? (uint32_t)G.eq_max_blocks : (mode1 ? 768u : 2048u);
  return want < 1 ? 1 : (want > cap ? cap : want);
}
static inline uint32_t sc_blocks_bind(uint32_t hq) {
  const uint32_t want = (hq + 255) / 256;
  return want < 1 ? 1 : (want > 4096 ? 4096 : want);
}
static constexpr size_t kScPartialBytes = 4096 * 128;  // per mailbox slot: 4096 blocks x up to 32 words

// the device heaps of eq tables of one EqSumCheckInstance (sumcheck.rs:608-641; scalars: ScAlg::Eq)
template <int FID> struct ScEqDev {
  using F = Fp<FID>;
  using H = HostFp4<FID>;
  uint32_t l = 0, first_half = 0, second_half = 0, KL = 0, KR = 0;
  uint32_t *heapL = nullptr, *heapR = nullptr;
  static size_t heap_bytes(uint32_t l_) {
    const uint32_t fh = l_ / 2, sh = l_ - fh, kl = fh > 0 ? fh - 1 : 0;
    return ((((size_t)2 << kl) * 32 + 255) & ~(size_t)255) + ((size_t)2 << sh) * 32 + 512;
  }
  // builds the tables on the stream (no wait) into [mem, mem + heap_bytes(l))
  void init(ScDev<FID>& h, const typename ScAlg<FID>::Eq& eq, char* mem) {
    l = eq.l, first_half = eq.first_half, second_half = eq.second_half;
    KL = first_half > 0 ? first_half - 1 : 0, KR = second_half;
    heapL = (uint32_t*)mem;
    heapR = (uint32_t*)(mem + ((((size_t)2 << KL) * 32 + 255) & ~(size_t)255));
    DeviceBackend be(h.c, false, false);
    uint32_t w[8] = {1, 0, 0, 0, 0, 0, 0, 0};  // ONE in the vectors' form
    if (h.mont) H::one().to_mont256(w);
    if (KL <= EqHeapFn<FID>::kMaxEll && KR <= EqHeapFn<FID>::kMaxEll) {
      EqHeapFn<FID> f;
      f.heapL = heapL, f.heapR = heapR, f.KL = KL, f.KR = KR, f.one = F::from_words(w);
      for (uint32_t i = 0; i < 2 * EqHeapFn<FID>::kMaxEll; i++) f.r[i] = f.nr[i] = F::zero();
      // left side: taus[1 .. first_half) (sumcheck.rs:634-635: skip(1)); right side: taus[first_half .. l)
      for (uint32_t i = 0; i < KL; i++) f.r[i] = eq.taus[1 + i].to_device(), f.nr[i] = eq.eq0[1 + i].to_device();
      for (uint32_t i = 0; i < KR; i++) f.r[KL + i] = eq.taus[first_half + i].to_device(), f.nr[KL + i] = eq.eq0[first_half + i].to_device();
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
          ScEqStepFn<FID> f{top, eq.taus[first_tau + (uint32_t)j].to_device(), size};
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
};

// the launches of one eq-factored instance (MODE 1 or 3) or of quad_prod (MODE 4) over in-place tables
template <int FID, int MODE> struct ScPass {
  using F = Fp<FID>;
  using H = HostFp4<FID>;
  using Tables = typename ScEqDev<FID>::Tables;
  ScDev<FID>& h;
  uint32_t *A, *B, *C;
  uint32_t* partial;  // device scratch of this instance (kScPartialBytes)
  uint32_t slot;      // mailbox slot of the sums; tail areas slot, slot + 1, slot + 2 take the tables
  hipStream_t stream; // the context's, or a side stream (batch prover: one per claim)
  F nk;
  static constexpr uint32_t NT = MODE == 3 ? 3u : MODE == 4 ? 2u : 1u;
  static constexpr bool kQuadForm = MODE == 3 || MODE == 4;  // the provers whose index is a chain worth spreading over four lanes
  static bool quad_on() { return G.sc_quad.load(std::memory_order_relaxed) != 0; }
  static bool host_parts_on() { return G.sc_host_parts.load(std::memory_order_relaxed) != 0; }
  ScPass(ScDev<FID>& h_, void* a, void* b, void* cc, uint32_t* partial_, uint32_t slot_)
      : h(h_), A((uint32_t*)a), B((uint32_t*)b), C((uint32_t*)cc), partial(partial_), slot(slot_), stream(h_.c.stream) {
    F fconst = F::zero();
    if (MODE == 3) {
      if (h.mont) fconst = pow2_plain<FID>(256);
      else fconst.l[0] = 1;
    }
    nk = MODE == 3 ? F::sub2(F::zero(), fconst.canon()).norm().canon() : F::zero();  // p - k, as eq_sums_t
  }
  static constexpr uint32_t kFactors = MODE == 3 ? 3u : 2u;  // stored factors per term without eqL
  uint32_t factors(const Tables& t) const { return kFactors + (MODE != 4 && t.eqL ? 1u : 0u); }
  // sums only over tables of `len` elements (round 1, and the high-half sum of the fallback)
  uint32_t sums(const uint32_t* a, const uint32_t* b, const uint32_t* cc, size_t len, const Tables& t) {
    const uint32_t hh = (uint32_t)(len / 2), seq = h.next_seq();
    hipStream_t s = stream;
    if (hh <= kScSmallHq) {
      ScSmallArgs<FID> x{a, b, cc, nullptr, nullptr, nullptr, t.eqL, t.eqR, F::zero(), nk, t.shift, t.mask, hh, 0u, seq, h.slot_dev(slot)};
      if (kQuadForm && hh <= 64 && quad_on()) hipLaunchKernelGGL((k_sc_small<FID, MODE, kQuadForm>), dim3(1), dim3(256), 0, s, x);
      else hipLaunchKernelGGL((k_sc_small<FID, MODE>), dim3(1), dim3(256), 0, s, x);
      h.launched();
    } else {
      uint32_t blocks = sc_blocks_sums(hh, MODE == 1);
      if (MODE == 4) {
        hipLaunchKernelGGL((k_plain_sums<FID, 1>), dim3(blocks), dim3(256), 0, s, a, b, (const uint32_t*)nullptr, hh, partial);
        hipLaunchKernelGGL((k_sum_partials_mail<FID, 2, 4>), dim3(1), dim3(256), 0, s, partial, blocks, h.slot_dev(slot), seq);
      } else {
        constexpr int M = MODE == 4 ? 1 : MODE;
        if (t.eqL && t.shift < 31) {
          // k_eq_rows walks whole rows of 2^shift indices: below 2^24 one row (or one block's worth of short rows) per block
          // fills more of the chip than the 8-indices-per-lane rule
          const uint32_t row = 1u << t.shift, rpb = row < 256u ? 256u / row : 1u, nrows = (uint32_t)(((uint64_t)hh + row - 1) >> t.shift);
          const uint32_t by_rows = (nrows + rpb - 1) / rpb;
          if (by_rows > blocks) blocks = by_rows < 2048u ? by_rows : 2048u;
          hipLaunchKernelGGL((k_eq_rows<FID, M>), dim3(blocks), dim3(256), 0, s, a, b, cc, t.eqL, t.eqR, t.shift, hh, nk, partial);
        } else {
          hipLaunchKernelGGL((k_eq_sums<FID, M>), dim3(blocks), dim3(256), 0, s, a, b, cc, t.eqL, t.eqR, t.shift, t.mask, hh, nk, partial);
        }
        hipLaunchKernelGGL((k_sum_partials_mail<FID, 2, 2>), dim3(1), dim3(256), 0, s, partial, blocks, h.slot_dev(slot), seq);
      }
      h.launched(2);
    }
    HIPCHK(hipGetLastError());
    return seq;
  }
  // one launch: bind + sums of a round (fused-sum forms).  Four lanes per index (sc_quad_loop) up to kScQuadMaxHq indices -- one
  // block up to 64 of them, 64 per block beyond --, else one lane per index: one block up to kScSmallHq, k_sc_pass beyond; passes of
  // <= kHostPartBlocks blocks leave the adding of their partials to the host.
  // returns the number of blocks whose partials the host is to add for this pass (0: the sums arrive added up)
  uint32_t launch_bind(const ScSmallArgs<FID>& x, uint32_t hq) {
    hipStream_t s = stream;
    uint32_t host_blocks = 0;
    const bool fused = G.sc_fused_sum.load(std::memory_order_relaxed) != 0;
    if (kQuadForm && quad_on() && (hq <= 64 || (fused && hq <= kScQuadMaxHq))) {
      if (hq <= 64) {
        hipLaunchKernelGGL((k_sc_small<FID, MODE, kQuadForm>), dim3(1), dim3(256), 0, s, x);
      } else {
        const uint32_t blocks = (hq + 63) / 64;
        ScPassArgs<FID> y{x, partial, partial + kScPartialBytes / 4 - 64};
        if (host_parts_on() && blocks <= kHostPartBlocks) y.host_part = h.part_dev(slot), host_blocks = blocks;
        hipLaunchKernelGGL((k_sc_pass<FID, MODE, kQuadForm>), dim3(blocks), dim3(256), 0, s, y);
      }
    } else if (hq <= kScSmallHq) {
      hipLaunchKernelGGL((k_sc_small<FID, MODE>), dim3(1), dim3(256), 0, s, x);
    } else {
      const uint32_t blocks = sc_blocks_bind(hq);
      ScPassArgs<FID> y{x, partial, partial + kScPartialBytes / 4 - 64};  // the ticket: a zero word at the end of this instance's scratch
      if (host_parts_on() && blocks <= kHostPartBlocks) y.host_part = h.part_dev(slot), host_blocks = blocks;
      hipLaunchKernelGGL((k_sc_pass<FID, MODE>), dim3(blocks), dim3(256), 0, s, y);
    }
    HIPCHK(hipGetLastError());
    h.launched();
    return host_blocks;
  }
  // bind the tables (len elements each) with r in place AND the sums of the next round over the bound halves
  uint32_t bind_sums(size_t len, const H& rh, const Tables& t) {
    const uint32_t hq = (uint32_t)(len / 4), seq = h.next_seq();
    const F r = rh.to_device();
    hipStream_t s = stream;
    const bool fused = G.sc_fused_sum.load(std::memory_order_relaxed) != 0;
    if (fused || hq <= kScSmallHq) {
      h.parts[slot] = launch_bind(ScSmallArgs<FID>{A, B, C, A, B, C, t.eqL, t.eqR, r, nk, t.shift, t.mask, hq, 1u, seq, h.slot_dev(slot)}, hq);
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
  // ---- a pass enqueued before its challenge exists.  Between two small passes lie ~13 us (timeline_2p20.txt): the sums reach the
  // host, algebra, transcript, LAUNCH, first wave -- the last two are 8-9 us on an idle queue.  Everything a bind + sums pass needs
  // except r is known a round ahead (tables, sizes, the eq tables of its round, its mailbox sequence): it is enqueued behind the
  // pass before it, starts the moment that one ends, and takes r from a line of pinned memory (sc_challenge); the host's "launch"
  // becomes one 48-byte write.  Only passes of at most kPrelaunchMaxHq indices (<= 64 polling blocks); never when the coming
  // round may need the fallback's extra pass on this stream (the caller checks Eq::l1p_zero) or when the caller asked for
  // synchronising waits (sc_poll_us = 0); every pre-launched pass is sent its challenge or cancelled, and one that hears nothing
  // leaves after 2 s.
  static constexpr uint32_t kPrelaunchMaxHq = 1u << 14;
  uint32_t pre_seq = 0, pre_parts = 0, pre_line = 0;  // mailbox sequence / host-added blocks / challenge line of the pass in flight (h.armed_seq[slot]: its challenge sequence)
  bool armed() const { return h.armed_seq[slot] != 0; }
  bool can_prelaunch(size_t len) {
    return G.sc_prelaunch.load(std::memory_order_relaxed) != 0 && G.sc_fused_sum.load(std::memory_order_relaxed) != 0 &&
           G.sc_poll_us.load(std::memory_order_relaxed) != 0 && len / 4 >= 1 && len / 4 <= kPrelaunchMaxHq && h.chal_line(slot) != nullptr;
  }
  // blocks launch_bind will start for a bound half of hq indices (the same rule, evaluated before the launch)
  uint32_t bind_blocks(uint32_t hq) const {
    const bool fused = G.sc_fused_sum.load(std::memory_order_relaxed) != 0;
    if (kQuadForm && quad_on() && (hq <= 64 || (fused && hq <= kScQuadMaxHq))) return hq <= 64 ? 1u : (hq + 63) / 64;
    return hq <= kScSmallHq ? 1u : sc_blocks_bind(hq);
  }
  // false: the device-wide budget of waiting blocks is spent -- the caller launches this pass when its challenge is known
  bool prelaunch(size_t len, const Tables& t) {
    if (!h.arm_reserve(slot, bind_blocks((uint32_t)(len / 4)))) return false;
    const uint32_t hq = (uint32_t)(len / 4), seq = h.next_seq(), cs = h.next_seq();
    ScSmallArgs<FID> x{A, B, C, A, B, C, t.eqL, t.eqR, F::zero(), nk, t.shift, t.mask, hq, 1u, seq, h.slot_dev(slot)};
    pre_line = h.take_line(slot) ? kChalLineWords : 0u;
    x.chal = h.chal_line(slot) + pre_line, x.chal_seq = cs;
    pre_parts = launch_bind(x, hq);
    pre_seq = seq;
    h.armed_seq[slot] = cs;
    h.armed_line[slot] = pre_line;
    return true;
  }
  // the challenge for the pass in flight; returns the mailbox sequence its sums will carry
  uint32_t send(const H& rh) {
    const F r = rh.to_device();
    ScDev<FID>::chal_write(h.chal_line(slot) + pre_line, r.l, 0u, h.armed_seq[slot]);
    h.armed_seq[slot] = 0;
    h.armed_line[slot] = 0;
    h.arm_release(slot);  // (the pass is running now; its blocks leave within microseconds)
    h.parts[slot] = pre_parts;  // (only now: until here the slot's pending result was the pass before)
    return pre_seq;
  }
  // ---- every remaining device round in ONE resident kernel (k_sc_resident; option sc_resident) -----------------------------------
  // Launched a round ahead like a pre-launched pass, from tables of <= kResidentMaxLen elements; the host's part of a round shrinks
  // to: add the blocks' partial sums, algebra, transcript, one 64-byte write.  res_next = the pass whose challenge goes out next
  // (== res_passes: the hand-over's).  h.armed_seq[slot] always holds the challenge sequence the kernel is waiting for, so
  // ScDev::cancel_armed (any failure, any fall-back) ends it.
  static constexpr bool kResQuad = kQuadForm;
  bool res_on = false;
  bool res_allowed = true;  // false: this pass shares its stream with other claims' passes (a resident kernel would hold them up for good)
  uint32_t res_seq0 = 0, res_cs0 = 0, res_next = 0, res_passes = 0, res_line0 = 0;
  uint32_t res_line(uint32_t i) const { return ((i + res_line0) & 1u) ? kChalLineWords : 0u; }  // word offset of pass i's line
  size_t res_len0 = 0;
  bool res_active() const { return res_on && h.armed_seq[slot] != 0; }
  bool can_resident(size_t len) {
    if (!res_allowed || !G.sc_resident.load(std::memory_order_relaxed) || len > kResidentMaxLen || len / 2 <= h.tail_len || len < 4) return false;
    if (kResQuad && !quad_on()) return false;  // (the four-lane form is the cubic / quad_prod provers' small-pass form; off: launched passes)
    return G.sc_fused_sum.load(std::memory_order_relaxed) != 0 && G.sc_poll_us.load(std::memory_order_relaxed) != 0 &&
           host_parts_on() && slot + NT <= kMailSlots && h.chal_line(slot) != nullptr;
  }
  // round0: the round whose challenge the first pass binds with.  eqd: the instance's eq heaps (MODE 4: nullptr)
  bool start_resident(size_t len, uint32_t round0, const ScEqDev<FID>* eqd) {
    uint32_t passes = 0;
    for (size_t l_ = len; l_ / 2 > h.tail_len; l_ /= 2) passes++;
    const uint32_t blocks = sc_res_blocks<FID, MODE, kResQuad>((uint32_t)(len / 4));
    if (passes == 0 || blocks > kHostPartBlocks || !h.arm_reserve(slot, blocks)) return false;
    ScResArgs<FID> a;
    a.A = A, a.B = B, a.C = C;
    a.heapL = eqd ? eqd->heapL : nullptr, a.heapR = eqd ? eqd->heapR : nullptr;
    a.first_half = eqd ? eqd->first_half : 0, a.second_half = eqd ? eqd->second_half : 0, a.l = eqd ? eqd->l : 0, a.round0 = round0;
    a.nk = nk;
    a.len0 = (uint32_t)len, a.passes = passes, a.tail = 1u;
    a.seq0 = res_seq0 = h.reserve_seq(passes + 1), a.cs0 = res_cs0 = h.reserve_seq(passes + 1);
    a.chal = h.chal_line(slot);
    a.line0 = res_line0 = h.take_line(slot);
    if (passes & 1u) (void)h.take_line(slot);  // passes + 1 waits in all, the hand-over's on line (line0 + passes) & 1: the next waiter takes the other one
    a.host_part = h.part_dev(slot);
    a.slot = h.slot_dev(slot);
    for (uint32_t t = 0; t < 3; t++) a.host_tab[t] = t < NT ? h.tail_dev(slot + t) : nullptr;
    a.ntab = NT;
    hipLaunchKernelGGL((k_sc_resident<FID, MODE, kResQuad>), dim3(blocks), dim3(256), 0, stream, a);
    HIPCHK(hipGetLastError());
    h.launched();
    res_on = true, res_next = 0, res_passes = passes, res_len0 = len;
    h.armed_seq[slot] = res_cs0;
    h.armed_line[slot] = res_line(0);
    return true;
  }
  // the challenge of the next bind + sums pass; returns the mailbox sequence its partial sums carry
  uint32_t res_send(const H& rh) {
    const F r = rh.to_device();
    const uint32_t i = res_next++;
    h.armed_seq[slot] = res_cs0 + i + 1;  // what the kernel waits for once this pass is through, and where: a cancel names that line
    h.armed_line[slot] = res_line(i + 1);
    ScDev<FID>::chal_write(h.chal_line(slot) + res_line(i), r.l, 0u, res_cs0 + i);
    h.parts[slot] = sc_res_blocks<FID, MODE, kResQuad>((uint32_t)((res_len0 >> i) / 4));
    return res_seq0 + i;
  }
  // the last challenge: the kernel binds once more and lands the tables of `half` elements in the tail areas
  void res_to_host(size_t half, const H& rh, std::vector<H>* out[3]) {
    require(res_next == res_passes && half == (res_len0 >> res_passes) / 2, NMX_E_HIP, "sum-check: resident kernel out of step");
    const F r = rh.to_device();
    h.armed_seq[slot] = 0;
    h.armed_line[slot] = 0;
    h.arm_release(slot);
    res_on = false;
    ScDev<FID>::chal_write(h.chal_line(slot) + res_line(res_passes), r.l, 0u, res_cs0 + res_passes);
    h.parts[slot] = 0;
    (void)h.wait(slot, res_seq0 + res_passes);
    for (uint32_t t = 0; t < NT; t++) {
      const uint32_t* src = h.tail_host(slot + t);
      out[t]->resize(half);
      for (size_t i = 0; i < half; i++) (*out[t])[i] = h.stored(src + 8 * i);
    }
  }
  ~ScPass() {
    if (armed()) h.cancel_armed();
  }
  // the hand-over: bind with r (rp == nullptr: no bind) and land the tables of `half` elements in the tail areas; fills out[]
  void to_host(size_t half, const H* rp, std::vector<H>* out[3]) {
    require(half <= kTailMax && slot + NT <= kMailSlots, NMX_E_HIP, "sum-check: tail hand-over out of range");
    ScBindOutArgs<FID> a;
    uint32_t* X[3] = {A, B, C};
    for (uint32_t t = 0; t < 3; t++) a.X[t] = t < NT ? X[t] : nullptr, a.host[t] = t < NT ? h.tail_dev(slot + t) : nullptr;
    a.r = rp ? rp->to_device() : F::zero();
    a.n = NT, a.half = (uint32_t)half, a.bind = rp ? 1u : 0u, a.seq = h.next_seq(), a.slot = h.slot_dev(slot);
    hipLaunchKernelGGL((k_sc_bind_to_host<FID>), dim3(1), dim3(256), 0, stream, a);
    HIPCHK(hipGetLastError());
    h.launched();
    (void)h.wait(slot, a.seq);
    for (uint32_t t = 0; t < NT; t++) {
      const uint32_t* src = h.tail_host(slot + t);
      out[t]->resize(half);
      for (size_t i = 0; i < half; i++) (*out[t])[i] = h.stored(src + 8 * i);
    }
  }
  // t(1) of the current round: the sums pass over a copy of the tables with the halves swapped (the fallback; never on
  // transcript-derived challenges)
  H high_half_sum(size_t len, const Tables& t) {
    // hipMalloc / hipFree below wait for the whole device: with a pass of this call waiting for ITS challenge they would never
    // return.  The callers never pre-launch into a round that takes this path (Eq::l1p_zero, checked over ALL claims of a batch).
    require(!h.any_armed(), NMX_E_HIP, "sum-check: the tau = 0 fall-back with a pre-launched pass in flight");
    const size_t hb = len / 2 * 32;
    char* tmp = nullptr;
    HIPCHK(hipMalloc((void**)&tmp, (size_t)NT * len * 32));
    const uint32_t* src[3] = {A, B, C};
    try {
      for (uint32_t i = 0; i < NT; i++) {
        HIPCHK(hipMemcpyAsync(tmp + (size_t)i * len * 32, (const char*)src[i] + hb, hb, hipMemcpyDeviceToDevice, stream));
        HIPCHK(hipMemcpyAsync(tmp + (size_t)i * len * 32 + hb, src[i], hb, hipMemcpyDeviceToDevice, stream));
      }
      const uint32_t* ta = (const uint32_t*)tmp;
      const uint32_t seq = sums(ta, NT > 1 ? ta + 8 * len : nullptr, NT > 2 ? ta + 16 * len : nullptr, len, t);
      const H v = h.raw(h.wait(slot, seq), factors(t));
      stream_wait(stream);
      (void)hipFree(tmp);
      return v;
    } catch (...) {
      (void)hipStreamSynchronize(stream);
      (void)hipFree(tmp);
      throw;
    }
  }
};

// SumcheckProof::prove_cubic_with_three_inputs (MODE 3, sumcheck.rs:446-507) / prove_quad_prod (MODE 4, sumcheck.rs:199-249)
template <int FID, int MODE>
static void sc_prove_t(Ctx& c, const void* claim, const void* taus, size_t num_rounds, void* A, void* B, void* C, uint32_t flags,
                       TranscriptFn cb, void* cb_ctx, uint8_t* out_polys, uint8_t* out_r, uint8_t* out_claims) {
  using H = HostFp4<FID>;
  const auto T0 = std::chrono::steady_clock::now();
  constexpr uint32_t NCO = MODE == 3 ? 4u : 3u, NT = MODE == 3 ? 3u : 2u;
  const uint32_t l = (uint32_t)num_rounds;
  try {
    ScDev<FID> h(c, flags);
    try {
    arena_reserve(c, kScPartialBytes + 512);
    HIPCHK(hipMemsetAsync(c.arena + kScPartialBytes - 256, 0, 256, c.stream));  // k_sc_pass's ticket
    typename ScAlg<FID>::Eq eq;
    ScEqDev<FID> eqd;
    size_t len = (size_t)1 << l;
    if (MODE == 3) {
      eq.init(h.alg, (const uint8_t*)taus, l);
      if (len > h.tail_len) {
        aux_reserve(c, ScEqDev<FID>::heap_bytes(l));
        eqd.init(h, eq, c.aux);
      }
    }
    ScPass<FID, MODE> pass(h, A, B, C, (uint32_t*)c.arena, 0);
    H cl = h.alg.in(claim);
    std::vector<H> hA, hB, hC;
    std::vector<H>* tabs[3] = {&hA, &hB, &hC};
    uint32_t j = 1;
    if (len <= h.tail_len) {
      pass.to_host(len, nullptr, tabs);  // the whole instance fits the tail
    } else {
      typename ScEqDev<FID>::Tables tb = MODE == 3 ? eqd.tables(1) : typename ScEqDev<FID>::Tables{nullptr, nullptr, 0, 0};
      uint32_t seq = pass.sums(pass.A, pass.B, pass.C, len, tb);
      if (MODE == 3) eq.prepare();  // the round's inversion runs under the pass (sc_host.hpp Eq::prepare)
      // the pass of round `next_round` (it binds tables of cur_len elements) goes out a round early when it can (ScPass::prelaunch)
      auto maybe_prelaunch = [&](size_t cur_len, uint32_t next_round) {
        if (pass.res_active()) return;  // the resident kernel has this pass (and all after it)
        if (MODE == 3 && eq.l1p_zero) return;  // the round in between takes the fallback: its extra pass needs this stream free
        if (pass.can_resident(cur_len) && pass.start_resident(cur_len, next_round - 1, MODE == 3 ? &eqd : nullptr)) return;
        if (cur_len / 2 <= h.tail_len || !pass.can_prelaunch(cur_len)) return;  // (the hand-over is never pre-launched)
        (void)pass.prelaunch(cur_len, MODE == 3 ? eqd.tables(next_round) : typename ScEqDev<FID>::Tables{nullptr, nullptr, 0, 0});
      };
      maybe_prelaunch(len, 2);
      for (;; j++) {
        const uint32_t* res = h.wait(0, seq);
        const H t0 = h.raw(res, pass.factors(tb)), t1 = h.raw(res + 8, pass.factors(tb));
        H co[4];
        if (MODE == 3) {
          H s0, lead, sm1;
          eq.derive(t0, t1, cl, false, s0, lead, sm1, [&] {
            h.cancel_armed();  // (a resident kernel waiting for the next challenge leaves: the fall-back's extra pass synchronises)
            return t0.dbl() + t1.dbl() - pass.high_half_sum(len, tb);
          });
          ScAlg<FID>::from_evals_deg3(s0, cl, lead, sm1, co);
        } else {
          ScAlg<FID>::from_evals_deg2(t0, cl, t1, co);
        }
        const H r = h.ask(cb, cb_ctx, co, NCO, out_polys ? out_polys + 32 * NCO * (size_t)(j - 1) : nullptr, out_r ? out_r + 32 * (size_t)(j - 1) : nullptr);
        cl = ScAlg<FID>::poly_eval(co, NCO, r);
        if (MODE == 3) eq.bound(r);
        h.prof.rounds++;
        if (len / 2 <= h.tail_len) {  // the bound tables go to the host: the remaining rounds (none if they are the final values) run there
          if (pass.res_active()) pass.res_to_host(len / 2, r, tabs);
          else pass.to_host(len / 2, &r, tabs);
          len /= 2;
          j++;
          break;
        }
        if (MODE == 3) tb = eqd.tables(j + 1);
        seq = pass.res_active() ? pass.res_send(r) : pass.armed() ? pass.send(r) : pass.bind_sums(len, r, tb);
        if (MODE == 3) eq.prepare();
        len /= 2;
        maybe_prelaunch(len, j + 2);
      }
    }
    if (j <= l) {
      h.prof.host_rounds += l - j + 1;
      sc_tail_rounds<FID, MODE>(h.alg, &eq, l, j, cl, hA, hB, hC, cb, cb_ctx, out_polys, out_r);
    }
    if (out_claims) {
      h.alg.out(hA[0], out_claims), h.alg.out(hB[0], out_claims + 32);
      if (NT == 3) h.alg.out(hC[0], out_claims + 64);
    }
    stream_wait(c.stream);  // the (partly bound) tables are the caller's again
    h.collect_torn_rejects();
    h.finish_profile(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - T0).count());
    } catch (...) {
      h.sync_all_quiet();  // whatever failed (a HIP call, a mailbox wait, a range check): no kernel of this call still writes A / B / C
      throw;
    }
  } catch (const ScFail& f) {
    rethrow(f);
  }
}

// SumcheckProof::prove_batch_eval (sumcheck.rs:251-353): k claims P_i(x_i) = e_i over polynomials of different sizes; the
// polynomials are bound in place (the reference binds clones, spartan/mod.rs:407-410: hand in copies to keep the originals).
// The round loop is sc_batch_rounds (sc_host.hpp); this supplies the device's side of it.
template <int FID> struct ScBatchDev {
  using H = HostFp4<FID>;
  ScDev<FID>& h;
  std::vector<ScBatchClaim<FID>>& claims;
  std::vector<ScPass<FID, 1>> pass;
  std::vector<ScEqDev<FID>> eqd;
  std::vector<size_t> len;
  std::vector<uint32_t> seq;
  std::vector<typename ScEqDev<FID>::Tables> tb;
  std::vector<H> last_t0;
  ScBatchDev(ScDev<FID>& h_, std::vector<ScBatchClaim<FID>>& c_) : h(h_), claims(c_) {}
  void start(size_t i) {
    tb[i] = eqd[i].tables(1);
    seq[i] = pass[i].sums(pass[i].A, nullptr, nullptr, len[i], tb[i]);
  }
  H t0(size_t i) { return last_t0[i] = h.raw(h.wait((uint32_t)i, seq[i]), pass[i].factors(tb[i])); }
  H t_m1(size_t i) {  // t(-1) = 2 t(0) - t(1)
    h.cancel_armed();  // every claim's waiting pass / resident kernel leaves: the fall-back's extra pass waits for the whole device
    return last_t0[i].dbl() - pass[i].high_half_sum(len[i], tb[i]);
  }
  void bind(size_t i, const H& r) {
    if (len[i] / 2 <= h.tail_len) {
      std::vector<H>* out[3] = {&claims[i].host, nullptr, nullptr};
      if (pass[i].res_active()) pass[i].res_to_host(len[i] / 2, r, out);
      else pass[i].to_host(len[i] / 2, &r, out);
    } else if (pass[i].res_active()) {  // the claim's resident kernel: the challenge is all a round needs
      seq[i] = pass[i].res_send(r);
      tb[i] = eqd[i].tables(claims[i].eq.round + 1);
    } else if (pass[i].armed()) {  // enqueued a round ago (ahead): all it lacks is r
      seq[i] = pass[i].send(r);
      tb[i] = tb_next[i];
    } else {
      tb[i] = eqd[i].tables(claims[i].eq.round + 1);
      seq[i] = pass[i].bind_sums(len[i], r, tb[i]);
    }
    len[i] /= 2;
  }
  // the pass that will bind claim i's table (len[i] elements now) goes out before its challenge exists (ScPass::prelaunch), unless it is
  // the hand-over or the coming round takes the fallback (whose extra pass needs the claim's stream)
  std::vector<typename ScEqDev<FID>::Tables> tb_next;
  void ahead(size_t i) {
    if (pass[i].res_active() || claims[i].eq.l1p_zero) return;
    if (pass[i].can_resident(len[i]) && pass[i].start_resident(len[i], claims[i].eq.round, &eqd[i])) return;
    if (len[i] / 2 <= h.tail_len || !pass[i].can_prelaunch(len[i])) return;
    if (tb_next.size() < pass.size()) tb_next.resize(pass.size());
    tb_next[i] = eqd[i].tables(claims[i].eq.round + 1);
    (void)pass[i].prelaunch(len[i], tb_next[i]);  // (false: the budget of waiting blocks is spent; bind() launches it late)
  }
};
template <int FID>
static void sc_prove_batch_t(Ctx& c, const uint8_t* claims_b, const size_t* num_rounds, void* const* polys, const uint8_t* const* eq_points,
                             const uint8_t* coeffs, size_t k, uint32_t flags, TranscriptFn cb, void* cb_ctx, uint8_t* out_polys,
                             uint8_t* out_r, uint8_t* out_finals) {
  using H = HostFp4<FID>;
  const auto T0 = std::chrono::steady_clock::now();
  require(k >= 1 && k <= kMailSlots, NMX_E_ARG, "prove_batch_eval: between 1 and 16 claims");
  try {
    ScDev<FID> h(c, flags);
    try {
    size_t heap_total = 0;
    for (size_t i = 0; i < k; i++) {
      require(num_rounds[i] >= 1 && num_rounds[i] < 31, NMX_E_ARG, "prove_batch_eval: 1 <= num_rounds < 31");
      heap_total += (ScEqDev<FID>::heap_bytes((uint32_t)num_rounds[i]) + 255) & ~(size_t)255;
    }
    arena_reserve(c, k * kScPartialBytes + 512);
    for (size_t i = 0; i < k; i++) HIPCHK(hipMemsetAsync(c.arena + (i + 1) * kScPartialBytes - 256, 0, 256, c.stream));  // the tickets
    aux_reserve(c, heap_total);
    std::vector<ScBatchClaim<FID>> cs(k);
    ScBatchDev<FID> dev(h, cs);
    dev.pass.reserve(k);
    dev.eqd.resize(k), dev.len.resize(k), dev.seq.assign(k, 0), dev.tb.resize(k), dev.last_t0.resize(k);
    size_t off = 0;
    for (size_t i = 0; i < k; i++) {
      cs[i].num_rounds = (uint32_t)num_rounds[i];
      cs[i].eq.init(h.alg, eq_points[i], cs[i].num_rounds);
      cs[i].claim0 = cs[i].running = h.alg.in(claims_b + 32 * i);
      cs[i].coeff = h.alg.in(coeffs + 32 * i);
      dev.len[i] = (size_t)1 << num_rounds[i];
      dev.pass.emplace_back(h, polys[i], nullptr, nullptr, (uint32_t*)(c.arena + i * kScPartialBytes), (uint32_t)i);
      if (dev.len[i] <= h.tail_len) {  // fits the tail as it is
        std::vector<H>* out[3] = {&cs[i].host, nullptr, nullptr};
        dev.pass[i].to_host(dev.len[i], nullptr, out);
      } else {
        dev.eqd[i].init(h, cs[i].eq, c.aux + off);
        off += (ScEqDev<FID>::heap_bytes((uint32_t)num_rounds[i]) + 255) & ~(size_t)255;
      }
    }
    // the claims of a round are independent passes over their own tables: claim i > 0 runs on side stream i - 1 (ordered behind
    // the set-up above), so a round costs one pass's latency, not k of them
    if (G.sc_side_streams.load(std::memory_order_relaxed))
      for (size_t i = 1; i < k; i++) dev.pass[i].stream = h.side_stream((uint32_t)(i - 1));
    // A kernel that stays for several rounds must never sit in front of a pass the round's challenge depends on.  With one claim that
    // cannot happen (the kernel's only counterpart is this thread).  With several it can even on separate HIP streams: the runtime
    // maps streams onto a handful of hardware queues, two "independent" streams may share one, and claim A's resident kernel then holds
    // up claim B's pass, whose sums the host needs before it can send A's challenge -- measured: every batch of four claims stalled
    // until the 4 s poll limit (gpurun_out/r6d).  Batches of several claims therefore keep one pass per round (pre-launched a round
    // ahead: passes are enqueued in dependency order, which is safe on a shared queue).
    if (k > 1)
      for (size_t i = 0; i < k; i++) dev.pass[i].res_allowed = false;
    sc_batch_rounds<FID>(h.alg, cs, dev, cb, cb_ctx, out_polys, out_r, out_finals, &h.prof.rounds);
    h.sync_all();
    h.collect_torn_rejects();
    h.finish_profile(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - T0).count());
    } catch (...) {
      h.sync_all_quiet();  // nothing of this call may still be running (context or side stream) when the tables go back to the caller
      throw;
    }
  } catch (const ScFail& f) {
    rethrow(f);
  }
}

// MultilinearPolynomial::evaluate_with / multi_evaluate_with (multilinear.rs:98-180) over HBM-resident polynomials, the provers' way:
// every polynomial's row pass and final sum are enqueued back to back (ScPass<1>::sums: k_eq_rows + one-block sum), the k results land
// in k mailbox slots and the host polls them -- no device-to-host copy, no stream synchronisation, no blocking wake-up per
// polynomial (round 5: eq tables, then per polynomial pass + sum + copy + synchronise: 0.10 ms for Cz(r_x), E(r_x) at 2^20 against
// ~25 us of kernels each).  eqL / eqR: the two sqrt-size tables (EvalScratch), in the vectors' form.
template <int FID>
static void mle_multi_eval_t(Ctx& c, const void* const* zs, size_t k, size_t len, const uint32_t* eqL, const uint32_t* eqR, uint32_t s_right,
                             uint32_t flags, uint8_t* out) {
  using H = HostFp4<FID>;
  ScDev<FID> h(c, flags);
  try {
    arena_reserve(c, k * kScPartialBytes + 512);
    const typename ScEqDev<FID>::Tables t{eqL, eqR, s_right, s_right >= 32 ? 0xffffffffu : ((1u << s_right) - 1u)};
    std::vector<ScPass<FID, 1>> pass;
    pass.reserve(k);
    std::vector<uint32_t> seq(k);
    const bool prof = G.profiling;
    DeviceBackend be(c, false, prof);
    be.mark("passes");
    for (size_t j = 0; j < k; j++) {
      pass.emplace_back(h, const_cast<void*>(zs[j]), nullptr, nullptr, (uint32_t*)(c.arena + j * kScPartialBytes), (uint32_t)j);
      seq[j] = pass[j].sums(pass[j].A, nullptr, nullptr, 2 * len, t);  // the mode-1 sum over "half" = len
    }
    be.mark("end");
    for (size_t j = 0; j < k; j++) {
      const H v = h.raw(h.wait((uint32_t)j, seq[j]), 3);  // three stored factors: z, eqL, eqR
      h.alg.out(v, out + 32 * j);
    }
    if (prof && be.nmarks == 2) {  // kernel time of the k passes + final sums (hipEvents on the call's stream), as the single-pass path reports it
      float ms = 0;
      HIPCHK(hipEventSynchronize(c.ev[1]));
      HIPCHK(hipEventElapsedTime(&ms, c.ev[0], c.ev[1]));
      prof_store(&ms, 1);
    }
  } catch (...) {
    h.sync_all_quiet();
    throw;
  }
}
void fv_mle_multi_eval(Ctx& c, int field, const void* const* zs, size_t k, size_t len, const uint32_t* eqL, const uint32_t* eqR,
                       uint32_t s_right, uint32_t flags, uint8_t* out) {
  require(k >= 1 && k <= kMailSlots, NMX_E_ARG, "mle_multi_eval: 1 .. 16 polynomials per pass");
  try {
    switch (field) {
      case 0: mle_multi_eval_t<0>(c, zs, k, len, eqL, eqR, s_right, flags, out); return;
      case 1: mle_multi_eval_t<1>(c, zs, k, len, eqL, eqR, s_right, flags, out); return;
      case 2: mle_multi_eval_t<2>(c, zs, k, len, eqL, eqR, s_right, flags, out); return;
      case 3: mle_multi_eval_t<3>(c, zs, k, len, eqL, eqR, s_right, flags, out); return;
      default: throw Fail{NMX_E_ARG, "bad field id"};
    }
  } catch (const ScFail& f) {
    rethrow(f);
  }
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
