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
template <int FID> __device__ __forceinline__ Fp<FID> quad_pick(const XYZZ<FID>& p, uint32_t q) {
  return q == 0 ? p.x : q == 1 ? p.y : q == 2 ? p.zz : p.zzz;
}

// dbl-2008-s-1 (curve.hpp dbl_in_place), three steps.  c = this lane's coordinate; returns the new one.
template <int FID> __device__ __forceinline__ Fp<FID> quad_dbl(const Fp<FID>& c, uint32_t q) {
  using F = Fp<FID>;
  if (qperm_u32<QP_L2>(c.is_zero_limbs() ? 1u : 0u)) return c;  // identity (quad-uniform)
  const F x = qperm<QP_L0>(c), y = qperm<QP_L1>(c);
  const F u = y.dbl().norm();                                   //  7.0
  const F m1 = fsel(q == 0, x, u);
  const F t1 = m1 * m1;                                         //  l0: xx < 1.23 ; others: v = u^2 < 1.39
  const F xx = qperm<QP_L0>(t1), v = qperm<QP_L1>(t1);
  const F m = (xx.dbl() + xx).norm();                           //  3.69
  const F a2 = q == 0 ? x : q == 1 ? u : q == 2 ? c : m;
  const F b2 = fsel(q == 3, m, v);
  const F t2 = a2 * b2;                                         //  l0: s = x*v ; l1: w = u*v ; l2: zz*v ; l3: m^2
  const F s = qperm<QP_L0>(t2), w = qperm<QP_L1>(t2), mm = qperm<QP_L3>(t2);
  const F x3 = F::sub4(mm, s.dbl().norm()).norm();              //  < 5.11
  const F e = F::sub8(s, x3).norm();                            //  < 9.06
  const F ny = F::sub4(F::zero(), y);                           //  4p - y, limbs < 2^30
  const F t3 = F::mul_add(fsel(q == 3, c, m), fsel(q == 3, w, e), w, fsel(q == 3, F::zero(), ny));
  //                                                                l1: y3 = m*e - w*y ; l3: zzz*w
  return q == 0 ? x3 : q == 2 ? t2 : t3;
}

// add-2008-s (curve.hpp add): (c1 coordinates) += (c2 coordinates); returns this lane's coordinate of the sum.
template <int FID> __device__ __forceinline__ Fp<FID> quad_add(const Fp<FID>& c1, const Fp<FID>& c2, uint32_t q) {
  using F = Fp<FID>;
  if (qperm_u32<QP_L2>(c2.is_zero_limbs() ? 1u : 0u)) return c1;  // += identity
  if (qperm_u32<QP_L2>(c1.is_zero_limbs() ? 1u : 0u)) return c2;  // identity += o
  const bool lo = q < 2;
  const F o2 = qperm<QP_SWAP2>(c2);
  const F t1 = c1 * o2;                      //  l0: u1 = x1*zz2 ; l1: s1 = y1*zzz2 ; l2: u2 = zz1*x2 ; l3: s2 = zzz1*y2
  const F tp = qperm<QP_SWAP2>(t1);
  const F a = fsel(lo, tp, t1);              //  the "2" product of this lane's pair (u2 | s2)
  const F b = fsel(lo, t1, tp);              //  the "1" product (u1 | s1)
  const F d = F::sub2(a, b).norm();          //  lanes 0,2: p = u2-u1 ; lanes 1,3: r = s2-s1      in (0.9, 3.1)
  if (qperm_u32<QP_L0>(d.maybe_zero_mod_p() ? 1u : 0u)) {
    // p may be 0 mod p (P == +-Q, or a 2^-29 false alarm): single-lane formulas on gathered operands
    XYZZ<FID> A = quad_gather<FID>(c1);
    A.add(quad_gather<FID>(c2));
    return quad_pick<FID>(A, q);
  }
  const F t2 = fsel(lo, d, c1) * fsel(lo, d, c2);   //  l0: pp ; l1: rr ; l2: zz1*zz2 ; l3: zzz1*zzz2
  const F pp = qperm<QP_L0>(t2);
  const F u1 = qperm<QP_L0>(b);
  const F m3 = q == 0 ? d : q == 1 ? u1 : t2;
  const F t3 = m3 * pp;                      //  l0: ppp = p*pp ; l1: q = u1*pp ; l2: zz3 ; l3: unused
  const F ppp = qperm<QP_L0>(t3), qq = qperm<QP_L1>(t3), rr = qperm<QP_L1>(t2);
  const F r = qperm<QP_L1>(d), s1 = qperm<QP_L1>(b);
  const F tt = (ppp + qq.dbl()).norm();      //  3.05
  const F x3 = F::sub4(rr, tt).norm();       //  < 5.08
  const F e = F::sub8(qq, x3).norm();        //  < 9.01
  const F ns1 = F::sub2(F::zero(), s1);      //  2p - s1, limbs < 2^30
  const F t4 = F::mul_add(fsel(q == 3, t2, r), fsel(q == 3, ppp, e), ppp, fsel(q == 3, F::zero(), ns1));
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
template <int FID> __device__ __forceinline__ Fp<FID> quad_madd(const Fp<FID>& c, const Fp<FID>& o, uint32_t q) {
  using F = Fp<FID>;
  if (qperm_u32<QP_L2>(c.is_zero_limbs() ? 1u : 0u)) return q < 2 ? o : F::one();  // identity += P  (msm.rs:133-139)
  const F t1 = o * c;                              //  l2: u2 < 1.01 ; l3: s2 < 1.02
  const F a = qperm<QP_SWAP2>(t1);                 //  l0: u2 ; l1: s2
  const F d = F::sub8(a, c).norm();                //  l0: p in (2.7, 9.01) ; l1: r < 9.02
  if (qperm_u32<QP_L0>(d.maybe_zero_mod_p() ? 1u : 0u)) {
    // P == +-Q, or a 2^-29 false alarm: single-lane formulas on gathered operands
    XYZZ<FID> A = quad_gather<FID>(c);
    A.add_affine(qperm<QP_L0>(o), qperm<QP_L1>(o));
    return quad_pick<FID>(A, q);
  }
  const F p = qperm<QP_L0>(d);
  const F t2 = fsel(q == 1, d, p).sqr();           //  l1: rr < 1.65 ; others: pp < 1.64
  const F t3 = fsel(q == 3, p, c) * t2;            //  l0: q < 1.07 ; l2: zz3 < 1.02 ; l3: ppp < 1.12
  const F rr = qperm<QP_L1>(t2), ppp = qperm<QP_L3>(t3), qq = qperm<QP_L0>(t3);
  const F tt = (ppp + qq.dbl()).norm();            //  3.26
  const F x3 = F::sub4(rr, tt).norm();             //  < 5.65 ... rr < 1.65: (1.65 + 4) within the x < 8 bound of sub8
  const F e = F::sub8(qq, x3).norm();              //  < 9.07
  const F ny = F::sub4(F::zero(), c);              //  l1: 4p - y1, limbs < 2^31
  const F t4 = F::mul_add(fsel(q == 3, c, d), fsel(q == 3, t3, e), ppp, fsel(q == 3, F::zero(), ny));
  //                                                   l1: y3 = r*e - y1*ppp ; l3: zzz3 = zzz1*ppp
  return q == 0 ? x3 : q == 2 ? t3 : t4;
}

}  // namespace nmx
#endif  // __HIPCC__

#if defined(__HIPCC__) || defined(__HIP__)
#include "msm_kernels.hpp"
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

}  // namespace nmx
#endif
