// curve_quad.hpp -- XYZZ addition / doubling computed by FOUR cooperating lanes (device only).
//
// The fold and bucket-reduction stages are chains of *dependent* point additions with far fewer work items than the
// chip has lanes (a 2^15-bucket tree ends in one pair): their cost is the latency of one addition, ~14 field
// multiplications back to back on one lane.  Here a quad (4 consecutive lanes) owns one point, lane k holding
// coordinate k of (X, Y, ZZ, ZZZ); the 12M + 2S of add-2008-s are scheduled as four steps in which every lane does
// ONE multiplication, operands moving between lanes with DPP quad permutes (v_mov_b32 ... quad_perm, full rate):
//     step 1   [ u1 = x1*zz2 | s1 = y1*zzz2 | u2 = zz1*x2 | s2 = zzz1*y2 ]
//     step 2   [ pp = p*p    | rr = r*r     | zz1*zz2     | zzz1*zzz2    ]        p = u2-u1, r = s2-s1
//     step 3   [ ppp = p*pp  | q = u1*pp    | zz3 = ..*pp |      -       ]
//     step 4   [     -       | y3 = r*e - s1*ppp (one reduction) | - | zzz3 = ..*ppp ]     x3 = rr-ppp-2q, e = q-x3
// 4 multiplication latencies instead of 13.5; the result is again distributed one coordinate per lane, so chains of
// additions never gather.  Same formulas, bounds and exceptional cases as curve.hpp (msm.rs:91-123, 65-88); the
// rare P == +-Q case falls back to the single-lane code on gathered operands.
#pragma once
#include "curve.hpp"

#if defined(__HIPCC__) || defined(__HIP__)
namespace nmx {

// quad_perm control words: lane i of every quad reads lane sel[i]
enum : int { QP_SWAP2 = 0x4E /* 2,3,0,1 */, QP_L0 = 0x00, QP_L1 = 0x55, QP_L2 = 0xAA, QP_L3 = 0xFF };

template <int CTRL> __device__ __forceinline__ uint32_t qperm_u32(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xF, 0xF, true);
}
template <int CTRL, int FID> __device__ __forceinline__ Fp<FID> qperm(const Fp<FID>& a) {
  Fp<FID> r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.l[i] = qperm_u32<CTRL>(a.l[i]);
  return r;
}
template <int FID> __device__ __forceinline__ Fp<FID> fsel(bool c, const Fp<FID>& a, const Fp<FID>& b) {
  Fp<FID> r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.l[i] = c ? a.l[i] : b.l[i];
  return r;
}

// coordinate `q` (= lane & 3) of a point at rest
template <int FID> __device__ __forceinline__ Fp<FID> quad_load(const XYZZW& m, uint32_t q) {
  return Fp<FID>::from_words(m.w + 8 * q);
}
template <int FID> __device__ __forceinline__ void quad_store(XYZZW& m, uint32_t q, const Fp<FID>& c) {
  c.canon().to_words(m.w + 8 * q);
}
// all four coordinates in every lane (slow path only)
template <int FID> __device__ __forceinline__ XYZZ<FID> quad_gather(const Fp<FID>& c) {
  XYZZ<FID> p;
  p.x = qperm<QP_L0>(c);
  p.y = qperm<QP_L1>(c);
  p.zz = qperm<QP_L2>(c);
  p.zzz = qperm<QP_L3>(c);
  return p;
}
// limb by limb: `q == 0 ? p.x : ...` on the structs selects an ADDRESS, which keeps the whole point in scratch memory
// (148 bytes of private segment in every quad kernel, found in round 4's ISA scan; only the rare P == +-Q path touched it: no
// measurable change in the tail's stage times, profiles/r04_msm_2p20/digit_ab.txt)
template <int FID> __device__ __forceinline__ Fp<FID> quad_pick(const XYZZ<FID>& p, uint32_t q) {
  Fp<FID> r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.l[i] = q == 0 ? p.x.l[i] : q == 1 ? p.y.l[i] : q == 2 ? p.zz.l[i] : p.zzz.l[i];
  return r;
}

// dbl-2008-s-1 (curve.hpp dbl_in_place), three steps.  c = this lane's coordinate; returns the new one.
template <int FID, bool LAT = kLatTail> __device__ __forceinline__ Fp<FID> quad_dbl(const Fp<FID>& c, uint32_t q) {
  using F = Fp<FID>;
  if (qperm_u32<QP_L2>(c.is_zero_limbs() ? 1u : 0u)) return c;  // identity (quad-uniform)
  const F x = qperm<QP_L0>(c), y = qperm<QP_L1>(c);
  const F u = y.dbl().norm();                                   //  7.0
  const F m1 = fsel(q == 0, x, u);
  const F t1 = F::template mulx<LAT>(m1, m1);                                         //  l0: xx < 1.23 ; others: v = u^2 < 1.39
  const F xx = qperm<QP_L0>(t1), v = qperm<QP_L1>(t1);
  const F m = (xx.dbl() + xx).norm();                           //  3.69
  const F a2 = q == 0 ? x : q == 1 ? u : q == 2 ? c : m;
  const F b2 = fsel(q == 3, m, v);
  const F t2 = F::template mulx<LAT>(a2, b2);                                         //  l0: s = x*v ; l1: w = u*v ; l2: zz*v ; l3: m^2
  const F s = qperm<QP_L0>(t2), w = qperm<QP_L1>(t2), mm = qperm<QP_L3>(t2);
  const F x3 = F::sub4(mm, s.dbl().norm()).norm();              //  < 5.11
  const F e = F::sub8(s, x3).norm();                            //  < 9.06
  const F ny = F::sub4(F::zero(), y);                           //  4p - y, limbs < 2^30
  const F t3 = F::template mul_addx<LAT>(fsel(q == 3, c, m), fsel(q == 3, w, e), w, fsel(q == 3, F::zero(), ny));
  //                                                                l1: y3 = m*e - w*y ; l3: zzz*w
  return q == 0 ? x3 : q == 2 ? t2 : t3;
}

// add-2008-s (curve.hpp add): (c1 coordinates) += (c2 coordinates); returns this lane's coordinate of the sum.
template <int FID, bool LAT = kLatTail> __device__ __forceinline__ Fp<FID> quad_add(const Fp<FID>& c1, const Fp<FID>& c2, uint32_t q) {
  using F = Fp<FID>;
  if (qperm_u32<QP_L2>(c2.is_zero_limbs() ? 1u : 0u)) return c1;  // += identity
  if (qperm_u32<QP_L2>(c1.is_zero_limbs() ? 1u : 0u)) return c2;  // identity += o
  const bool lo = q < 2;
  const F o2 = qperm<QP_SWAP2>(c2);
  const F t1 = F::template mulx<LAT>(c1, o2);                      //  l0: u1 = x1*zz2 ; l1: s1 = y1*zzz2 ; l2: u2 = zz1*x2 ; l3: s2 = zzz1*y2
  const F tp = qperm<QP_SWAP2>(t1);
  const F a = fsel(lo, tp, t1);              //  the "2" product of this lane's pair (u2 | s2)
  const F b = fsel(lo, t1, tp);              //  the "1" product (u1 | s1)
  const F d = F::sub2(a, b).norm();          //  lanes 0,2: p = u2-u1 ; lanes 1,3: r = s2-s1      in (0.9, 3.1)
  if (qperm_u32<QP_L0>(d.maybe_zero_mod_p() ? 1u : 0u)) {
    // p may be 0 mod p (P == +-Q, or a 2^-29 false alarm): single-lane formulas on gathered operands
    XYZZ<FID> A = quad_gather<FID>(c1);
    A.template add<LAT>(quad_gather<FID>(c2));
    return quad_pick<FID>(A, q);
  }
  const F t2 = F::template mulx<LAT>(fsel(lo, d, c1), fsel(lo, d, c2));   //  l0: pp ; l1: rr ; l2: zz1*zz2 ; l3: zzz1*zzz2
  const F pp = qperm<QP_L0>(t2);
  const F u1 = qperm<QP_L0>(b);
  const F m3 = q == 0 ? d : q == 1 ? u1 : t2;
  const F t3 = F::template mulx<LAT>(m3, pp);                      //  l0: ppp = p*pp ; l1: q = u1*pp ; l2: zz3 ; l3: unused
  const F ppp = qperm<QP_L0>(t3), qq = qperm<QP_L1>(t3), rr = qperm<QP_L1>(t2);
  const F r = qperm<QP_L1>(d), s1 = qperm<QP_L1>(b);
  const F tt = (ppp + qq.dbl()).norm();      //  3.05
  const F x3 = F::sub4(rr, tt).norm();       //  < 5.08
  const F e = F::sub8(qq, x3).norm();        //  < 9.01
  const F ns1 = F::sub2(F::zero(), s1);      //  2p - s1, limbs < 2^30
  const F t4 = F::template mul_addx<LAT>(fsel(q == 3, t2, r), fsel(q == 3, ppp, e), ppp, fsel(q == 3, F::zero(), ns1));
  //                                             l1: y3 = r*e - s1*ppp ; l3: zzz3 = zzz1*zzz2*ppp
  return q == 0 ? x3 : q == 2 ? t3 : t4;
}

// madd-2008-s (curve.hpp add_affine): (c coordinates) += the affine point whose halves the quad holds in `o`
// (even lanes: x canonical; odd lanes: y < 2p normalized, sign already applied).  Four multiplication steps:
//     step 1   [      -      |      -       | u2 = x2*zz1 | s2 = y2*zzz1 ]
//     step 2   [ pp = p*p    | rr = r*r     | pp          | pp           ]        p = u2-x1, r = s2-y1
//     step 3   [ q = x1*pp   |      -       | zz3 = zz1*pp| ppp = p*pp   ]
//     step 4   [      -      | y3 = r*e - y1*ppp (one reduction) | - | zzz3 = zzz1*ppp ]   x3 = rr-ppp-2q, e = q-x3
// against 9 dependent multiplications on one lane.  r carries +8p instead of curve.hpp's +4p (one subtraction
// constant for both lanes): y3 < p * (1 + (9.02*9.07 + 4.5)/127) < 1.7p, inside the y < 3.5p invariant.
template <int FID, bool LAT = kLatTail> __device__ __forceinline__ Fp<FID> quad_madd(const Fp<FID>& c, const Fp<FID>& o, uint32_t q) {
  using F = Fp<FID>;
  if (qperm_u32<QP_L2>(c.is_zero_limbs() ? 1u : 0u)) return q < 2 ? o : F::one();  // identity += P  (msm.rs:133-139)
  const F t1 = F::template mulx<LAT>(o, c);                              //  l2: u2 < 1.01 ; l3: s2 < 1.02
  const F a = qperm<QP_SWAP2>(t1);                 //  l0: u2 ; l1: s2
  const F d = F::sub8(a, c).norm();                //  l0: p in (2.7, 9.01) ; l1: r < 9.02
  if (qperm_u32<QP_L0>(d.maybe_zero_mod_p() ? 1u : 0u)) {
    // P == +-Q, or a 2^-29 false alarm: single-lane formulas on gathered operands
    XYZZ<FID> A = quad_gather<FID>(c);
    A.template add_affine<LAT>(qperm<QP_L0>(o), qperm<QP_L1>(o));
    return quad_pick<FID>(A, q);
  }
  const F p = qperm<QP_L0>(d);
  const F t2 = F::template sqrx<LAT>(fsel(q == 1, d, p));           //  l1: rr < 1.65 ; others: pp < 1.64
  const F t3 = F::template mulx<LAT>(fsel(q == 3, p, c), t2);            //  l0: q < 1.07 ; l2: zz3 < 1.02 ; l3: ppp < 1.12
  const F rr = qperm<QP_L1>(t2), ppp = qperm<QP_L3>(t3), qq = qperm<QP_L0>(t3);
  const F tt = (ppp + qq.dbl()).norm();            //  3.26
  const F x3 = F::sub4(rr, tt).norm();             //  < 5.65 ... rr < 1.65: (1.65 + 4) within the x < 8 bound of sub8
  const F e = F::sub8(qq, x3).norm();              //  < 9.07
  const F ny = F::sub4(F::zero(), c);              //  l1: 4p - y1, limbs < 2^31
  const F t4 = F::template mul_addx<LAT>(fsel(q == 3, c, d), fsel(q == 3, t3, e), ppp, fsel(q == 3, F::zero(), ny));
  //                                                   l1: y3 = r*e - y1*ppp ; l3: zzz3 = zzz1*ppp
  return q == 0 ? x3 : q == 2 ? t3 : t4;
}

}  // namespace nmx
#endif  // __HIPCC__

#if defined(__HIPCC__) || defined(__HIP__)
#include "msm_kernels.hpp"
#include "msm_partition.hpp"
#include "msm_seg.hpp"
namespace nmx {

// coordinate `q` of a raw-limb record (msm_seg.hpp)
template <int FID> __device__ __forceinline__ Fp<FID> quad_load_raw(const XYZZL& m, uint32_t q) {
  Fp<FID> r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.l[i] = m.l[9 * q + i];
  return r;
}
template <int FID> __device__ __forceinline__ void quad_store_raw(XYZZL& m, uint32_t q, const Fp<FID>& c) {
#pragma unroll
  for (int i = 0; i < 9; i++) m.l[9 * q + i] = c.l[i];
}
// FoldRawFn / FinalSegFn (msm_seg.hpp) with one quad per work item
template <int FID> struct FoldRawQuadFn {
  const uint32_t* counters;
  const HeavyRec* list;
  XYZZL* partial_raw;
  uint32_t T, cap, groups, use_big;
  __device__ __forceinline__ void operator()(uint32_t tid) const {
    if (use_big && counters[3] <= T) return;
    const uint32_t q = tid & 3u, item = tid >> 2;
    const uint32_t j = item % T, nh = counters[use_big ? 4 : 1];
    for (uint32_t h = item / T; h < nh; h += groups) {
      const HeavyRec r = list[h];
      const uint32_t cnt = r.cnt < cap ? r.cnt : cap;
      if (j + T >= cnt) continue;
      Fp<FID> acc = quad_load_raw<FID>(partial_raw[r.off + j], q);
      for (uint32_t k = j + T; k < cnt; k += T) acc = quad_add<FID>(acc, quad_load_raw<FID>(partial_raw[r.off + k], q), q);
      quad_store_raw<FID>(partial_raw[r.off + j], q, acc);
    }
  }
};
template <int FID> struct FinalSegQuadFn {
  const uint32_t* start;
  const uint32_t* end;
  const uint32_t* total_p;
  const XYZZL* bucket_raw;
  const XYZZL* partial_raw;
  XYZZW* buckets;
  uint32_t nbuckets, lanes, min_seg, heavy_above;
  __device__ __forceinline__ void operator()(uint32_t tid) const {
    const uint32_t q = tid & 3u, k = tid >> 2;
    if (k >= nbuckets) return;  // quad-uniform
    const uint32_t s0 = start[k], e0 = end[k];
    Fp<FID> acc = Fp<FID>::zero();  // zz = 0: the identity
    if (e0 > s0) {
      const uint32_t seg = seg_len(*total_p, lanes, min_seg);
      const uint32_t l0 = s0 / seg, l1 = (e0 - 1) / seg;
      uint32_t cnt = l1 - l0;
      if (cnt > SegPlan::kBigAbove) return;  // written by k_big_all (quad-uniform)
      if (cnt > heavy_above) cnt = heavy_above;
      acc = quad_load_raw<FID>(bucket_raw[k], q);
      for (uint32_t j = 0; j < cnt; j++) acc = quad_add<FID>(acc, quad_load_raw<FID>(partial_raw[l0 + 1 + j], q), q);
    }
    quad_store<FID>(buckets[k], q, acc);
  }
};

// FoldFn / ReducePairFn (msm_kernels.hpp) with one quad per work item.  tid = 4 * item + coordinate.
template <int FID> struct FoldQuadFn {
  const uint32_t* counters;
  const HeavyRec* heavy;
  XYZZW* partials;
  XYZZW* buckets;
  uint32_t T, cap, groups;
  __device__ __forceinline__ void operator()(uint32_t tid) const {
    if (T >= 64 && counters[3] <= T) return;
    const uint32_t q = tid & 3u, item = tid >> 2;
    const uint32_t j = item % T, nh = counters[T >= 64 ? 4 : 1];
    for (uint32_t h = item / T; h < nh; h += groups) {
      const HeavyRec r = heavy[h];
      const uint32_t cnt = r.cnt < cap ? r.cnt : cap;
      if (j >= cnt) continue;
      if (T != 1 && j + T >= cnt) continue;
      Fp<FID> acc = quad_load<FID>(partials[r.off + j], q);
      for (uint32_t k = j + T; k < cnt; k += T) acc = quad_add<FID>(acc, quad_load<FID>(partials[r.off + k], q), q);
      quad_store<FID>(T == 1 ? buckets[r.bucket] : partials[r.off + j], q, acc);
    }
  }
};

// AccumFn (msm_kernels.hpp) with one quad per task: for MSMs whose task count is below the chip's lane count the
// accumulate kernel is a chain of `lmax` dependent mixed additions per lane, i.e. latency-bound.
template <int FID> struct AccumQuadFn {
  const AffineW* bases;
  const uint32_t* vals;
  const uint32_t* start;
  const uint32_t* end;
  const uint32_t* counters;
  const TaskRec* extra;
  XYZZW* buckets;
  XYZZW* partials;
  MsmShape sh;

  // this lane's half of base v (even lanes x, odd lanes y with the digit's sign applied)
  __device__ __forceinline__ Fp<FID> half(uint32_t v, uint32_t q) const {
    using F = Fp<FID>;
    const F h = F::from_words(bases[v & 0x7fffffffu].w + 8 * (q & 1u));
    const F nh = F::sub2(F::zero(), h).norm();  // 2p - y
    return fsel((q & 1u) && (v >> 31), nh, h);
  }
  __device__ __forceinline__ Fp<FID> run(uint32_t b, uint32_t len, uint32_t q) const {
    using F = Fp<FID>;
    F acc = F::zero();
    if (len == 0) return acc;
    F cur = half(vals[b], q);
    for (uint32_t j = 0; j < len; j++) {
      F nxt = cur;
      if (j + 1 < len) nxt = half(vals[b + j + 1], q);  // in flight during the addition
      acc = quad_madd<FID>(acc, cur, q);
      cur = nxt;
    }
    return acc;
  }
  __device__ __forceinline__ void operator()(uint32_t tid) const {
    const uint32_t q = tid & 3u, t = tid >> 2;
    if (t < sh.nbuckets) {
      const uint32_t b = start[t], s = end[t] - b;
      if (s <= sh.lmax) quad_store<FID>(buckets[t], q, run(b, s, q));  // split buckets are written by the folds
    } else {
      const uint32_t e = t - sh.nbuckets;
      if (e >= counters[0]) return;
      const TaskRec r = extra[e];
      quad_store<FID>(partials[e], q, run(r.start, r.len, q));
    }
  }
};

template <int FID> struct ReducePairQuadFn {
  const XYZZW* D;
  const XYZZW* Y;
  XYZZW* D_out;
  XYZZW* Y_out;
  uint32_t n_in, pairs, pairs_padded /* multiple of 16: roles never share a wave */, first;
  __device__ __forceinline__ void operator()(uint32_t tid) const {
    const uint32_t q = tid & 3u, item = tid >> 2;
    const uint32_t role = item >= pairs_padded ? 1u : 0u;
    const uint32_t j = item - role * pairs_padded;
    if (j >= pairs) return;
    const uint32_t half = n_in >> 1;
    const uint32_t w = j / half, u = j - w * half;
    const size_t base = (size_t)w * n_in + 2 * (size_t)u;
    const size_t o = (size_t)w * half + u;
    if (role == 0) {
      if (n_in == 2) return;
      Fp<FID> d = quad_add<FID>(quad_load<FID>(D[base], q), quad_load<FID>(D[base + 1], q), q);
      quad_store<FID>(D_out[o], q, quad_dbl<FID>(d, q));
    } else {
      Fp<FID> y;
      if (first) {
        y = quad_add<FID>(quad_dbl<FID>(quad_load<FID>(D[base + 1], q), q), quad_load<FID>(D[base], q), q);
      } else {
        y = quad_add<FID>(quad_load<FID>(Y[base + 1], q), quad_load<FID>(D[base + 1], q), q);
        y = quad_add<FID>(y, quad_load<FID>(Y[base], q), q);
      }
      quad_store<FID>(Y_out[o], q, y);
    }
  }
};

// ----------------------------------------------------------------------------------------------------
// Block-level kernels of the MSM tail (round 3).  Round 2's tail was ~30 dependent launches of a few waves each (one per
// reduction level, five strided fold passes that exit at once on ordinary inputs): on some boxes of the pool every such
// launch cost twice what it did on others, and the driver-timed MSM lost 0.2 ms to it.  Here the dependency between
// steps is a __syncthreads() with the points in LDS (raw limbs, 144 B per point, stride-9 words per lane: conflict-free).
// ----------------------------------------------------------------------------------------------------
template <int FID> __device__ __forceinline__ Fp<FID> lds_load_pt(const uint32_t* buf, uint32_t idx, uint32_t q) {
  Fp<FID> r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.l[i] = buf[idx * 36u + q * 9u + i];
  return r;
}
template <int FID> __device__ __forceinline__ void lds_store_pt(uint32_t* buf, uint32_t idx, uint32_t q, const Fp<FID>& c) {
#pragma unroll
  for (int i = 0; i < 9; i++) buf[idx * 36u + q * 9u + i] = c.l[i];
}

// Sum of the block's per-quad points (NQ quads, a power of two), result in quad 0.  Every thread of the block calls it.
template <int FID, uint32_t NQ>
__device__ __forceinline__ Fp<FID> block_sum_quads(Fp<FID> acc, uint32_t* lds /* NQ * 36 words */, uint32_t qd, uint32_t q) {
  lds_store_pt<FID>(lds, qd, q, acc);
  __syncthreads();
  for (uint32_t st = NQ / 2; st >= 1; st >>= 1) {
    if (qd < st) acc = quad_add<FID>(acc, lds_load_pt<FID>(lds, qd + st, q), q);
    __syncthreads();
    if (qd < st) lds_store_pt<FID>(lds, qd, q, acc);
    __syncthreads();
  }
  return acc;
}

// Every big bucket (more than 64 continuation pieces; SegPlan) completely, in ONE launch whatever its size.  The plan step
// numbered the (bucket, slice) ITEMS (SegPlan: counters[6], rec.pad, big_items); a fixed grid of small blocks (128 threads = 32
// quads, four blocks per CU) walks them with the grid's stride.  An item: 32 quads x (slice / 32) strided additions, a 5-level
// LDS tree.  A bucket of one slice is then finished (+ bucket_raw[k], canonical store).  Otherwise the slice sum is parked in the
// slice's own first position and the block takes a ticket of its GROUP of 32 slices; the block that draws a group's last ticket
// sums the group's parked slices (one per quad, the same tree), parks that in the group's first position and takes a ticket of
// the bucket; the block that draws the bucket's last ticket sums the group sums and finishes.  No block ever waits for another.
// What it costs is the dependent chain: slice / 32 additions + three 5-level trees, whatever the input:
//   0/1 scalars (ONE bucket of 196 608 pieces), all-equal scalars (15 buckets of 26 214), 10-bit scalars (1023 buckets of 127).
// (Round 3's first versions used 512-thread blocks on 4096-piece slices, 16 blocks per possible slice index plus one per possible
// bucket, every block searching the list for its work: 0.2 ms for each of these inputs, most of it idle blocks scanning and
// seven-level trees with one block per CU -- and shorter slices, i.e. more parallel work, made it slower.)
struct BigAllArgs {
  const uint32_t* counters;
  const HeavyRec* big;
  const uint32_t* items;
  const uint32_t* gbase;
  const XYZZL* bucket_raw;
  XYZZL* partial_raw;
  XYZZW* buckets;
  uint32_t* done;   // [big capacity], zero-initialised: finished groups per bucket
  uint32_t* gdone;  // [group capacity], zero-initialised: finished slices per group
  uint32_t slice;   // pieces per item
};
template <int FID, int THREADS> __global__ __launch_bounds__(THREADS) void k_big_all(BigAllArgs a) {
  const uint32_t nitems = a.counters[6];
  constexpr uint32_t NQ = THREADS / 4, GR = SegPlan::kBigGroup;
  static_assert(GR <= NQ, "a group's parked slices are summed one per quad");
  __shared__ uint32_t lds[NQ * 36];
  __shared__ uint32_t s_ticket;
  const uint32_t qd = threadIdx.x >> 2, q = threadIdx.x & 3u;
  // parks `v` (held by quad 0) at `pos`, takes a ticket of `ctr`: true in the block that drew ticket `want` (block-uniform)
  // (the fences are not what the hand-over costs: a build without them, and one with agent-coherent sc1 accesses instead of
  // the L2 write-back, timed the same -- profiles/r03_msm_2p20/big_bucket_pass.txt)
  auto park_and_ticket = [&](const Fp<FID>& v, uint32_t pos, uint32_t* ctr, uint32_t want) {
    if (qd == 0) {
      quad_store_raw<FID>(a.partial_raw[pos], q, v);
      __threadfence();
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      __threadfence();
      s_ticket = atomicAdd(ctr, 1u);
    }
    __syncthreads();
    const bool last = s_ticket == want;
    __syncthreads();  // s_ticket is rewritten by the next ticket of this block
    if (last) __threadfence();
    return last;
  };
  auto load_parked = [&](uint32_t pos) { return quad_load_raw<FID>(a.partial_raw[pos], q); };
  for (uint32_t item = blockIdx.x; item < nitems; item += gridDim.x) {  // block-uniform
    const uint32_t h = a.items[item];
    const HeavyRec r = a.big[h];
    const uint32_t nsl = (r.cnt + a.slice - 1) / a.slice, s = item - r.pad;
    const uint32_t lo = s * a.slice, hi = r.cnt < lo + a.slice ? r.cnt : lo + a.slice;
    Fp<FID> acc = Fp<FID>::zero();  // zz = 0: the identity
    for (uint32_t p = lo + qd; p < hi; p += NQ) acc = quad_add<FID>(acc, quad_load_raw<FID>(a.partial_raw[r.off + p], q), q);
    acc = block_sum_quads<FID, NQ>(acc, lds, qd, q);
    if (nsl > 1) {
      const uint32_t ng = (nsl + GR - 1) / GR, g = s / GR, g_lo = g * GR, g_n = nsl - g_lo < GR ? nsl - g_lo : GR;
      if (!park_and_ticket(acc, r.off + lo, &a.gdone[a.gbase[h] + g], g_n - 1)) continue;
      acc = qd < g_n ? load_parked(r.off + (g_lo + qd) * a.slice) : Fp<FID>::zero();
      acc = block_sum_quads<FID, NQ>(acc, lds, qd, q);
      if (ng > 1) {
        if (!park_and_ticket(acc, r.off + g_lo * a.slice, &a.done[h], ng - 1)) continue;
        acc = Fp<FID>::zero();
        for (uint32_t t = qd; t < ng; t += NQ) acc = quad_add<FID>(acc, load_parked(r.off + t * GR * a.slice), q);
        acc = block_sum_quads<FID, NQ>(acc, lds, qd, q);
      }
    }
    if (qd == 0) {
      acc = quad_add<FID>(quad_load_raw<FID>(a.bucket_raw[r.bucket], q), acc, q);
      quad_store<FID>(a.buckets[r.bucket], q, acc);
    }
    __syncthreads();  // lds is reused by the next item
  }
}

// LV levels of the bucket-reduction pair tree (ReducePairFn, msm_kernels.hpp: D' = 2 (D_2j + D_2j+1),
// Y' = Y_2j + Y_2j+1 + D_2j+1) inside one block: a block owns S = THREADS / 4 consecutive inputs (the tree is oblivious to
// where a bucket set ends as long as 2^LV divides the set size), half of its quads compute D', the other half Y'
// (whole waves per role: the two formulas never share a wave), levels are separated by __syncthreads() with the points
// in LDS.  16 levels = 3 launches (6 + 5 + 5) instead of 16.
struct ReduceTreeArgs {
  const XYZZW* D;
  const XYZZW* Y;  // == D when first
  XYZZW* D_out;
  XYZZW* Y_out;
  uint32_t n_total;  // inputs of this launch (a multiple of 2^levels)
  uint32_t levels;   // 1 .. log2(S)
  uint32_t first;    // Y = D = B on entry
  uint32_t last;     // the tree ends with this launch: the final D is not needed
  const uint32_t* err_src;  // last launch: the pipeline's error word, copied behind the sums (one device->host copy)
};
template <int FID, int THREADS> __global__ __launch_bounds__(THREADS) void k_reduce_tree(ReduceTreeArgs a) {
  constexpr uint32_t S = THREADS / 4, H = S / 2;
  __shared__ uint32_t lds[(S + S / 2) * 36];
  // two buffers (levels alternate): [0] holds H points per role, [1] S / 4
  auto bufD = [&](uint32_t b) { return lds + b * (S * 36); };
  auto bufY = [&](uint32_t b) { return lds + b * (S * 36) + (b ? (S / 4) * 36 : H * 36); };
  // Roles alternate wave by wave (even waves D', odd waves Y'), output j lives in wave pair j / 16: the waves that are still
  // busy at the sparse levels -- the first of each role -- sit on different SIMDs.  (Roles by half-block put wave 0 and wave
  // THREADS / 128 on the same SIMD: a single wave of quad additions already fills 80 % of a SIMD's issue slots
  // (profiles/r03_msm_2p20/add_latency.txt), so the two roles ran one after the other: ~10 us per level instead of ~5.5.)
  const uint32_t q = threadIdx.x & 3u, wave = threadIdx.x >> 6;
  const uint32_t role = wave & 1u, j = (wave >> 1) * 16u + ((threadIdx.x & 63u) >> 2);
  const uint32_t base = blockIdx.x * S;
  const uint32_t n_in = a.n_total - base < S ? a.n_total - base : S;
  if (a.err_src && blockIdx.x == 0 && threadIdx.x == 0) *(uint32_t*)(a.Y_out + (a.n_total >> a.levels)) = *a.err_src;
  for (uint32_t lv = 1; lv <= a.levels; lv++) {
    const bool from_g = lv == 1, to_g = lv == a.levels;
    const uint32_t rb = lv & 1u, wb = (lv - 1u) & 1u;
    if (j < (n_in >> lv) && !(role == 0 && to_g && a.last)) {
      auto ldD = [&](uint32_t i) { return from_g ? quad_load<FID>(a.D[base + i], q) : lds_load_pt<FID>(bufD(rb), i, q); };
      auto ldY = [&](uint32_t i) { return from_g ? quad_load<FID>(a.Y[base + i], q) : lds_load_pt<FID>(bufY(rb), i, q); };
      const bool fst = from_g && a.first;
      const Fp<FID> d1 = ldD(2 * j + 1);
      Fp<FID> c;
      if (role == 0) {
        c = quad_dbl<FID>(quad_add<FID>(ldD(2 * j), d1, q), q);
      } else {
        // first level of the whole tree (Y = D = B):  B_2j + 2 B_2j+1 ; afterwards  (Y_2j+1 + D_2j+1) + Y_2j
        const Fp<FID> o1 = fst ? d1 : ldY(2 * j + 1), o0 = fst ? ldD(2 * j) : ldY(2 * j);
        c = fst ? quad_dbl<FID>(d1, q) : quad_add<FID>(o1, d1, q);
        c = quad_add<FID>(c, o0, q);
      }
      if (to_g) quad_store<FID>((role ? a.Y_out : a.D_out)[(base >> a.levels) + j], q, c);
      else lds_store_pt<FID>(role ? bufY(wb) : bufD(wb), j, q, c);
    }
    __syncthreads();
  }
}

// ----------------------------------------------------------------------------------------------------
// Small MSMs (round 5): bucket sums in TWO launches.  An MSM over a key of fewer than 2^14 points runs on c = 8 tables: 128
// buckets of ~W n / 128 entries each (10 538 pairs: 2 634 per bucket).  The task path spent nine launches on them -- plan,
// expand, accumulate (8-entry tasks), six strided fold passes of which three find work -- for a chain of 8 + 5 + 8 + 8
// dependent additions: 0.135 of the 0.25 ms such an MSM takes (profiles/r05_small_msm/stages_before.txt), which is what
// prove_step's secondary-curve commitments cost.  Here a BLOCK of 64 quads takes `chunk` consecutive entries of ONE bucket
// (the partition left every bucket's entries contiguous: start[k] .. end[k]), each quad adds its chunk / 64 entries and an
// one partial per quad comes out; a second launch, one block per BUCKET, adds the bucket's partials up (64 quads striding over
// them, then a six-level LDS tree: block_sum_quads).  Which (bucket, chunk) a block owns follows from the bucket sizes alone, so every block derives it from start[]
// / end[] itself (a scan over <= 1024 buckets in LDS): no plan kernel, no lists.  A bucket that collects everything (0 / 1
// or all-equal scalars) simply gets more blocks, and the second launch walks its partials 64 at a time.
// Chain: chunk / 64 mixed additions, then blocks-of-the-bucket + 6 additions (identity operands skip the arithmetic).
// ----------------------------------------------------------------------------------------------------
static constexpr uint32_t kSmallMaxBuckets = 1024;
struct SmallAccArgs {
  const AffineW* bases;
  const uint32_t* vals;
  const uint32_t* start;
  const uint32_t* end;
  XYZZW* part;     // 64 partials per block of k_small_accum (one per quad)
  XYZZW* buckets;  // k_small_combine's output
  uint32_t nbuckets, chunk;
};
// first block of every bucket into s_off[0 .. nbuckets] (s_off[nbuckets] = all blocks); every thread of a 256-thread block calls it
__device__ __forceinline__ void small_plan(const SmallAccArgs& a, uint32_t* s_off, uint32_t* s_nb /* 256 */, uint32_t* s_sc /* 257 */,
                                           uint32_t* s_w /* 16 */) {
  const uint32_t t = threadIdx.x, M = a.nbuckets, per = (M + 255u) / 256u;  // <= 4 buckets per thread
  uint32_t loc[4], sum = 0;
#pragma unroll
  for (uint32_t i = 0; i < 4; i++) {
    const uint32_t k = t * per + i;
    uint32_t nb = 0;
    if (i < per && k < M) nb = (a.end[k] - a.start[k] + a.chunk - 1u) / a.chunk;
    loc[i] = nb;
    sum += nb;
  }
  s_nb[t] = sum;
  __syncthreads();
  block_excl_scan(s_nb, s_sc, 256u, s_w);
  uint32_t run = s_sc[t];
#pragma unroll
  for (uint32_t i = 0; i < 4; i++) {
    const uint32_t k = t * per + i;
    if (i < per && k < M) s_off[k] = run;
    run += loc[i];
  }
  if (t == 0) s_off[M] = s_sc[256];
  __syncthreads();
}
template <int FID> __global__ __launch_bounds__(256) void k_small_accum(SmallAccArgs a) {
  using F = Fp<FID>;
  __shared__ uint32_t s_off[kSmallMaxBuckets + 1], s_nb[256], s_sc[257], s_w[16];
  small_plan(a, s_off, s_nb, s_sc, s_w);
  const uint32_t b = blockIdx.x;
  if (b >= s_off[a.nbuckets]) return;  // block-uniform
  const uint32_t k = find_bin(s_off, a.nbuckets, b), j = b - s_off[k];
  const uint32_t e_lo = a.start[k] + j * a.chunk, e_end = a.end[k];
  const uint32_t len = e_end - e_lo < a.chunk ? e_end - e_lo : a.chunk;
  const uint32_t qd = threadIdx.x >> 2, q = threadIdx.x & 3u;
  const uint32_t q_lo = e_lo + (uint32_t)((uint64_t)len * qd / 64u), q_hi = e_lo + (uint32_t)((uint64_t)len * (qd + 1u) / 64u);
  // this lane's half of an entry's base (even lanes x, odd lanes y with the digit's sign applied): AccumQuadFn::half
  auto half = [&](uint32_t v) {
    const F h = F::from_words(a.bases[v & 0x7fffffffu].w + 8 * (q & 1u));
    const F nh = F::sub2(F::zero(), h).norm();
    return fsel((q & 1u) && (v >> 31), nh, h);
  };
  F acc = F::zero();  // zz = 0: the identity
  if (q_hi > q_lo) {
    F cur = half(a.vals[q_lo]);
    for (uint32_t e = q_lo; e < q_hi; e++) {
      F nxt = cur;
      if (e + 1 < q_hi) nxt = half(a.vals[e + 1]);  // in flight during the addition
      acc = quad_madd<FID>(acc, cur, q);
      cur = nxt;
    }
  }
  // one partial per QUAD, no tree here: joining the 64 quad sums inside this block (a six-level LDS tree) kept three blocks per
  // CU resident through levels in which one wave works -- accumulate 90 us instead of 45 (profiles/r05_small_msm/stages.txt);
  // the second launch, with one block per BUCKET, does the joining
  quad_store<FID>(a.part[(size_t)b * 64u + qd], q, acc);
}
template <int FID> __global__ __launch_bounds__(256) void k_small_combine(SmallAccArgs a) {
  using F = Fp<FID>;
  __shared__ uint32_t s_off[kSmallMaxBuckets + 1], s_nb[256], s_sc[257], s_w[16];
  __shared__ uint32_t lds[64 * 36];
  small_plan(a, s_off, s_nb, s_sc, s_w);
  const uint32_t k = blockIdx.x, base = s_off[k] * 64u, n = (s_off[k + 1] - s_off[k]) * 64u;  // 64 partials per block of the bucket
  const uint32_t qd = threadIdx.x >> 2, q = threadIdx.x & 3u;
  F acc = F::zero();
  for (uint32_t p = qd; p < n; p += 64u) acc = quad_add<FID>(acc, quad_load<FID>(a.part[(size_t)base + p], q), q);
  acc = block_sum_quads<FID, 64>(acc, lds, qd, q);
  if (qd == 0) quad_store<FID>(a.buckets[k], q, acc);
}

}  // namespace nmx
#endif
