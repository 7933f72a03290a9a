// curve.hpp -- extended-Jacobian (X, Y, ZZ, ZZZ) bucket arithmetic for a = 0 short-Weierstrass curves.
//
// Same coordinate system and EFD formula set the reference uses for its bucket loops
// (/root/reference/src/provider/msm.rs:38-165: madd-2008-s, add-2008-s, dbl-2008-s-1), including its explicit
// handling of the exceptional cases (empty bucket, P == Q -> double, P == -Q -> empty, identity base skipped:
// msm.rs:92-113,130-155).  The formulas never use the curve constant b, so one instantiation per base field
// serves both curves of a cycle.
//
// Lazy-reduction discipline (fp.hpp): no conditional subtraction of p anywhere in the hot formulas.  In-register
// invariants of an XYZZ accumulator, in multiples of p (asserted in the NMX_DEBUG_BOUNDS emulation build):
//     x < 5.3      y < 3.5      zz, zzz < 1.2         every coordinate normalized
// A product of operands bounded by a*p and b*p is < (1 + a*b/127) p; the bound of every intermediate is written
// at the right of its line and stays below the 127 limit.  Points at rest in HBM are canonical (< p), packed
// 8 x u32 per coordinate.
#pragma once
#include "fp.hpp"

namespace nmx {

// 64 bytes in HBM: x || y, each the canonical internal-form residue packed in 8 x u32; identity = all zero
// (like halo2curves / traits.rs:303-312; (0, 0) is not on any of the four curves).
struct alignas(16) AffineW {  // 16-byte alignment: one point is four global_load_dwordx4
  uint32_t w[16];
};
// 128 bytes in HBM: x, y, zz, zzz packed; identity <=> zz words all zero (msm.rs:59-61)
struct alignas(16) XYZZW {
  uint32_t w[32];
};

// 144 bytes in HBM: the four coordinates as their 9 in-register limbs, NOT reduced -- whatever weakly reduced value the
// accumulator holds (bounds of the XYZZ invariants).  Storing it is 36 plain word stores and no arithmetic, which is
// what a lane of the segment-balanced accumulate kernel can afford inside a divergent branch; identity <=> zz limbs 0.
struct alignas(16) XYZZL {
  uint32_t l[36];
};

template <int FID> struct Affine {
  Fp<FID> x, y;  // canonical
  static NMX_HD Affine load(const AffineW& m) {
    Affine a;
    a.x = Fp<FID>::from_words(m.w);
    a.y = Fp<FID>::from_words(m.w + 8);
    return a;
  }
  NMX_HD void store(AffineW& m) const {
    x.to_words(m.w);
    y.to_words(m.w + 8);
  }
  NMX_HD bool is_identity() const { return x.is_zero_limbs() && y.is_zero_limbs(); }
};

template <int FID> struct XYZZ {
  using F = Fp<FID>;
  F x, y, zz, zzz;

  static NMX_HD XYZZ identity() {
    XYZZ r;
    r.x = F::one();
    r.y = F::one();
    r.zz = F::zero();
    r.zzz = F::zero();
    return r;
  }
  // Identity <=> zz is the literal zero: zz is only ever a product of non-zero residues, the constant one, or an
  // explicit zero written by identity() -- and 0 * anything comes out of the Montgomery product as literal 0.
  NMX_HD bool is_identity() const { return zz.is_zero_limbs(); }

  static NMX_HD XYZZ load(const XYZZW& m) {
    XYZZ r;
    r.x = F::from_words(m.w);
    r.y = F::from_words(m.w + 8);
    r.zz = F::from_words(m.w + 16);
    r.zzz = F::from_words(m.w + 24);
    return r;
  }
  NMX_HD void store(XYZZW& m) const {  // canonicalises: at rest every coordinate is < p
    x.canon().to_words(m.w);
    y.canon().to_words(m.w + 8);
    zz.canon().to_words(m.w + 16);
    zzz.canon().to_words(m.w + 24);
  }
  static NMX_HD XYZZ load_raw(const XYZZL& m) {
    XYZZ r;
#pragma unroll
    for (int i = 0; i < 9; i++) {
      r.x.l[i] = m.l[i];
      r.y.l[i] = m.l[9 + i];
      r.zz.l[i] = m.l[18 + i];
      r.zzz.l[i] = m.l[27 + i];
    }
    return r;
  }
  NMX_HD void store_raw(XYZZL& m) const {  // limbs must be normalized (they are after every add / dbl)
#pragma unroll
    for (int i = 0; i < 9; i++) {
      m.l[i] = x.l[i];
      m.l[9 + i] = y.l[i];
      m.l[18 + i] = zz.l[i];
      m.l[27 + i] = zzz.l[i];
    }
  }
  NMX_HD void check() const {
    x.check_below(5.3, "x");
    y.check_below(3.5, "y");
    zz.check_below(1.2, "zz");
    zzz.check_below(1.2, "zzz");
  }

  static NMX_HD XYZZ from_affine(const Affine<FID>& p) {
    if (p.is_identity()) return identity();
    XYZZ r;
    r.x = p.x;
    r.y = p.y;
    r.zz = F::one();
    r.zzz = F::one();
    return r;
  }

  NMX_HD XYZZ neg() const {  // rare (host tail)
    XYZZ r = *this;
    r.y = F::sub2(F::zero(), y.canon()).norm();  // 2p - y in (p, 2p]
    return r;
  }

  // dbl-2008-s-1 with a = 0 (msm.rs:65-88).                                     bound (x p)
  // LAT (here and below): the latency-oriented products of fp.hpp (kernels that run one or two waves per SIMD)
  template <bool LAT = false> NMX_HD void dbl_in_place() {
    if (is_identity()) return;
    // a point of order 2 (y == 0) cannot occur on these prime-order curves
    F u = y.dbl().norm();                          //  7.0
    F v = F::template sqrx<LAT>(u);                                 //  1 + 49/127      < 1.39
    F w = F::template mulx<LAT>(u, v);                                   //  1 + 9.8/127     < 1.08
    F s = F::template mulx<LAT>(x, v);                                   //  1 + 7.4/127     < 1.06
    F xx = F::template sqrx<LAT>(x);                                //  1 + 28.1/127    < 1.23
    F m = (xx.dbl() + xx).norm();                  //  3.69
    F s2 = s.dbl().norm();                         //  2.12
    F x3 = F::sub4(F::template sqrx<LAT>(m), s2).norm();            //  1.11 + 4        < 5.11
    F e = F::sub8(s, x3).norm();                   //  1.06 + 8        < 9.06
    F ny = F::sub4(F::zero(), y);                  //  4p - y in (0.5, 4], limbs < 2^31 (left un-normalized)
    F y3 = F::template mul_addx<LAT>(m, e, w, ny);                //  m*e - w*y:  1 + (33.5 + 4.4)/127 < 1.3   [one reduction]
    x = x3;
    y = y3;
    zz = F::template mulx<LAT>(zz, v);                                   //  1 + 1.67/127    < 1.02
    zzz = F::template mulx<LAT>(zzz, w);                                 //  < 1.02
#ifdef NMX_BOUND_CHECKS
    check();
#endif
  }

  // madd-2008-s (msm.rs:129-165): this += (px, +-py), the affine operand non-identity, px and py canonical (< p).
  // The sign of a signed window digit is applied to the PRODUCT: s2 = +-(py * zzz), i.e. r = s2 - y is formed either as
  // s2p + (4p - y) or as (2p - s2p) + (4p - y) -- 9 subtractions and 9 selects instead of negating, normalising and
  // selecting the operand (round 3: -24 instructions per addition); t = ppp + 2q enters x3 un-normalised against a spread
  // 4p whose limbs dominate 3 * 2^29 (-24 more).
  template <bool LAT = false> NMX_HD void add_affine_signed(const F& px, const F& py, bool negate) {
    if (is_identity()) {
      x = px;
      y = negate ? F::sub2(F::zero(), py).norm() : py;  // 2p - y in (p, 2p]
      zz = F::one();
      zzz = F::one();
      return;
    }
    F u2 = F::template mulx<LAT>(px, zz);          //  1 + 1.2/127     < 1.01
    F s2p = F::template mulx<LAT>(py, zzz);        //  1 + 1.2/127     < 1.01
    F d = F::sub8(u2, x).norm();                   //  in (2.7, 9.01)             [x < 5.3 < 8]
    if (d.maybe_zero_mod_p()) {                    //  taken with probability 2^-29 unless u2 == x
      if (F::eq_mod_p(u2, x)) {
        const F s2 = negate ? F::sub2(F::zero(), s2p).norm() : s2p;
        if (F::eq_mod_p(s2, y))
          dbl_in_place<LAT>();                     //  P == Q   (msm.rs:148-150)
        else
          *this = identity();                      //  P == -Q  (msm.rs:151-153)
        return;
      }
    }
    F r;                                           //  s2 - y + 4p < 5.02   |   -s2 - y + 6p in (1.4, 6)
#pragma unroll
    for (int i = 0; i < 9; i++) {
      const uint32_t t4 = F::PP::K4P[i] - y.l[i];                          //  [y < 3.5 < 4: K4P dominates its limbs]
      r.l[i] = (negate ? F::PP::K2P[i] - s2p.l[i] : s2p.l[i]) + t4;        //  [s2p < 2p - 2^233: K2P dominates]
    }
    r = r.norm();
    F pp = F::template sqrx<LAT>(d);               //  1 + 81.2/127    < 1.64
    F ppp = F::template mulx<LAT>(d, pp);          //  1 + 14.8/127    < 1.12
    F q = F::template mulx<LAT>(x, pp);            //  1 + 8.7/127     < 1.07
    F rr = F::template sqrx<LAT>(r);               //  1 + 36/127      < 1.29
    F x3;                                          //  rr - (ppp + 2q) + 4p  < 5.3: limbs of ppp + 2q are <= 3 (2^29 - 1), which the
#pragma unroll                                     //  spread 4p below dominates limb by limb (top limb: 3.26 p < 4p - 3)
    for (int i = 0; i < 9; i++) {
      const uint32_t k4w = F::PP::P4[i] + (i < 8 ? 3u << 29 : 0u) - (i > 0 ? 3u : 0u);
      x3.l[i] = rr.l[i] + (k4w - (ppp.l[i] + 2u * q.l[i]));
    }
    x3 = x3.norm();
    F e = F::sub8(q, x3).norm();                   //  1.07 + 8        < 9.07
    F ny = F::sub4(F::zero(), y);                  //  4p - y in (0.5, 4], limbs < 2^31 (left un-normalized)
    F y3 = F::template mul_addx<LAT>(r, e, ppp, ny);  //  r*e - y*ppp:  1 + (54.4 + 4.5)/127 < 1.47  [one reduction]
    x = x3;
    y = y3;
    zz = F::template mulx<LAT>(zz, pp);            //  1 + 1.97/127    < 1.02
    zzz = F::template mulx<LAT>(zzz, ppp);         //  < 1.02
#ifdef NMX_BOUND_CHECKS
    for (int i = 0; i < 9; i++) {
      const uint32_t k4w = F::PP::P4[i] + (i < 8 ? 3u << 29 : 0u) - (i > 0 ? 3u : 0u);
      if (ppp.l[i] + 2u * q.l[i] > k4w || s2p.l[i] > F::PP::K2P[i]) {
        fprintf(stderr, "add_affine_signed: limb %d not dominated\n", i);
        abort();
      }
    }
    check();
#endif
  }
  // madd-2008-s (msm.rs:129-165): this += (px, py), the affine operand non-identity.
  // px canonical (< p); py < 2p normalized (a canonical y, or 2p - y for a negated point).
  template <bool LAT = false> NMX_HD void add_affine(const F& px, const F& py) {
    if (is_identity()) {
      x = px;
      y = py;
      zz = F::one();
      zzz = F::one();
      return;
    }
    F u2 = F::template mulx<LAT>(px, zz);                                //  1 + 1.2/127     < 1.01
    F s2 = F::template mulx<LAT>(py, zzz);                               //  1 + 2.4/127     < 1.02
    F d = F::sub8(u2, x).norm();                   //  in (2.7, 9.01)             [x < 5.3 < 8]
    if (d.maybe_zero_mod_p()) {                    //  taken with probability 2^-29 unless u2 == x
      if (F::eq_mod_p(u2, x)) {
        if (F::eq_mod_p(s2, y))
          dbl_in_place<LAT>();                          //  P == Q   (msm.rs:148-150)
        else
          *this = identity();                      //  P == -Q  (msm.rs:151-153)
        return;
      }
    }
    F r = F::sub4(s2, y).norm();                   //  1.02 + 4        < 5.02     [y < 3.5 < 4]
    F pp = F::template sqrx<LAT>(d);                                //  1 + 81.2/127    < 1.64
    F ppp = F::template mulx<LAT>(d, pp);                                //  1 + 14.8/127    < 1.12
    F q = F::template mulx<LAT>(x, pp);                                  //  1 + 8.7/127     < 1.07
    F t = (ppp + q.dbl()).norm();                  //  3.26
    F x3 = F::sub4(F::template sqrx<LAT>(r), t).norm();             //  (1 + 25.2/127) + 4 < 5.2
    F e = F::sub8(q, x3).norm();                   //  1.07 + 8        < 9.07
    F ny = F::sub4(F::zero(), y);                  //  4p - y in (0.5, 4], limbs < 2^31 (left un-normalized)
    F y3 = F::template mul_addx<LAT>(r, e, ppp, ny);              //  r*e - y*ppp:  1 + (45.6 + 4.5)/127 < 1.4  [one reduction]
    x = x3;
    y = y3;
    zz = F::template mulx<LAT>(zz, pp);                                  //  1 + 1.97/127    < 1.02
    zzz = F::template mulx<LAT>(zzz, ppp);                               //  < 1.02
#ifdef NMX_BOUND_CHECKS
    check();
#endif
  }
  // affine operand as loaded from HBM; negate = the sign of a signed window digit
  // The sign is applied with a per-limb select, NOT a branch: lanes of one wave carry digits of both signs, and
  // two call sites of the (fully inlined) addition would make every wave execute it twice.
  template <bool LAT = false> NMX_HD void add_affine(const Affine<FID>& p, bool negate = false) {
    if (p.is_identity()) return;  // msm.rs:130-132
#ifndef NMX_MADD_R2
    add_affine_signed<LAT>(p.x, p.y, negate);
    return;
#endif
    F ny = F::sub2(F::zero(), p.y).norm();  // 2p - y in (p, 2p]
    F y;
#pragma unroll
    for (int i = 0; i < 9; i++) y.l[i] = negate ? ny.l[i] : p.y.l[i];
    add_affine<LAT>(p.x, y);
  }

  // add-2008-s (msm.rs:91-123): this += o, both operands within the in-register invariants.
  template <bool LAT = false> NMX_HD void add(const XYZZ& o) {
    if (o.is_identity()) return;
    if (is_identity()) {
      *this = o;
      return;
    }
    F u1 = F::template mulx<LAT>(x, o.zz);                               //  1 + 6.4/127     < 1.06
    F u2 = F::template mulx<LAT>(o.x, zz);                               //  < 1.06
    F s1 = F::template mulx<LAT>(y, o.zzz);                              //  1 + 4.2/127     < 1.04
    F s2 = F::template mulx<LAT>(o.y, zzz);                              //  < 1.04
    F d = F::sub2(u2, u1).norm();                  //  in (0.94, 3.06)
    if (d.maybe_zero_mod_p()) {
      if (F::eq_mod_p(u1, u2)) {
        if (F::eq_mod_p(s1, s2))
          dbl_in_place<LAT>();                          //  msm.rs:106-108
        else
          *this = identity();                      //  msm.rs:109-111
        return;
      }
    }
    F r = F::sub2(s2, s1).norm();                  //  3.04
    F pp = F::template sqrx<LAT>(d);                                //  1 + 9.4/127     < 1.08
    F ppp = F::template mulx<LAT>(d, pp);                                //  < 1.03
    F q = F::template mulx<LAT>(u1, pp);                                 //  < 1.01
    F t = (ppp + q.dbl()).norm();                  //  3.05
    F x3 = F::sub4(F::template sqrx<LAT>(r), t).norm();             //  (1 + 9.3/127) + 4 < 5.08
    F e = F::sub8(q, x3).norm();                   //  < 9.01
    F ns1 = F::sub2(F::zero(), s1);                //  2p - s1 in (0.9, 2], limbs < 2^31 (left un-normalized)
    F y3 = F::template mul_addx<LAT>(r, e, ppp, ns1);             //  r*e - s1*ppp:  1 + (27.4 + 2.1)/127 < 1.24 [one reduction]
    x = x3;
    y = y3;
    zz = F::template mulx<LAT>(F::template mulx<LAT>(zz, o.zz), pp);                         //  < 1.02
    zzz = F::template mulx<LAT>(F::template mulx<LAT>(zzz, o.zzz), ppp);                     //  < 1.02
#ifdef NMX_BOUND_CHECKS
    check();
#endif
  }

  // affine (x/zz, y/zzz) canonical; identity -> (0, 0)   (msm.rs:172-183 + traits.rs:303-312)
  NMX_HD Affine<FID> to_affine() const {
    Affine<FID> r;
    if (is_identity()) {
      r.x = F::zero();
      r.y = F::zero();
      return r;
    }
    // one inversion: (zz*zzz)^-1, then zz^-1 = inv*zzz, zzz^-1 = inv*zz
    F i = (zz * zzz).inv();
    r.x = (x * (i * zzz)).canon();
    r.y = (y * (i * zz)).canon();
    return r;
  }
};

}  // namespace nmx
