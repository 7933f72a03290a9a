// curve.hpp -- extended-Jacobian (X, Y, ZZ, ZZZ) bucket arithmetic for a = 0 short-Weierstrass curves.
//
// Same coordinate system and EFD formula set the reference uses for its bucket loops
// (/root/reference/src/provider/msm.rs:38-165: madd-2008-s, add-2008-s, dbl-2008-s-1), including its
// explicit handling of the exceptional cases (empty bucket, P == Q -> double, P == -Q -> empty, identity
// base skipped: msm.rs:92-113,130-155).  The formulas never use the curve constant b, so one instantiation
// per base field serves both curves of a cycle.
#pragma once
#include "fp.hpp"

namespace nmx {

template <int FID> struct Affine {  // 64 bytes; identity encoded as (0, 0) like halo2curves / traits.rs:303-312
  Fp<FID> x, y;
  NMX_HD bool is_identity() const { return x.is_zero() && y.is_zero(); }
};

template <int FID> struct XYZZ {  // 128 bytes; identity <=> zz == 0 (msm.rs:59-61)
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
  NMX_HD bool is_identity() const { return zz.is_zero(); }

  static NMX_HD XYZZ from_affine(const Affine<FID>& p) {
    if (p.is_identity()) return identity();
    XYZZ r;
    r.x = p.x;
    r.y = p.y;
    r.zz = F::one();
    r.zzz = F::one();
    return r;
  }

  NMX_HD XYZZ neg() const {
    XYZZ r = *this;
    r.y = y.neg();
    return r;
  }

  // dbl-2008-s-1 with a = 0: 2M + 5S ... (msm.rs:65-88)
  NMX_HD void dbl_in_place() {
    if (is_identity()) return;
    // a point of order 2 (y == 0) cannot occur on these prime-order curves
    F u = y.dbl();
    F v = u.sqr();
    F w = u * v;
    F s = x * v;
    F xx = x.sqr();
    F m = xx.dbl() + xx;
    F x3 = m.sqr() - s.dbl();
    F y3 = m * (s - x3) - w * y;
    x = x3;
    y = y3;
    zz = zz * v;
    zzz = zzz * w;
  }

  // madd-2008-s: this += (px, +-py), affine operand known non-identity by the caller when skip_check
  NMX_HD void add_affine(const F& px, const F& py) {
    if (is_identity()) {
      x = px;
      y = py;
      zz = F::one();
      zzz = F::one();
      return;
    }
    F u2 = px * zz;
    F s2 = py * zzz;
    if (u2 == x) {
      if (s2 == y) {
        dbl_in_place();
      } else {
        *this = identity();
      }
      return;
    }
    F p = u2 - x;
    F r = s2 - y;
    F pp = p.sqr();
    F ppp = p * pp;
    F q = x * pp;
    F x3 = r.sqr() - ppp - q.dbl();
    y = r * (q - x3) - y * ppp;
    x = x3;
    zz = zz * pp;
    zzz = zzz * ppp;
  }
  NMX_HD void add_affine(const Affine<FID>& p) {
    if (p.is_identity()) return;  // msm.rs:130-132
    add_affine(p.x, p.y);
  }

  // add-2008-s: this += o  (msm.rs:91-123)
  NMX_HD void add(const XYZZ& o) {
    if (o.is_identity()) return;
    if (is_identity()) {
      *this = o;
      return;
    }
    F u1 = x * o.zz;
    F u2 = o.x * zz;
    F s1 = y * o.zzz;
    F s2 = o.y * zzz;
    if (u1 == u2) {
      if (s1 == s2) {
        dbl_in_place();
      } else {
        *this = identity();
      }
      return;
    }
    F p = u2 - u1;
    F r = s2 - s1;
    F pp = p.sqr();
    F ppp = p * pp;
    F q = u1 * pp;
    F x3 = r.sqr() - ppp - q.dbl();
    y = r * (q - x3) - s1 * ppp;
    x = x3;
    zz = zz * o.zz * pp;
    zzz = zzz * o.zzz * ppp;
  }

  // affine (x/zz, y/zzz); identity -> (0, 0)   (msm.rs:172-183 + traits.rs:303-312)
  NMX_HD Affine<FID> to_affine() const {
    Affine<FID> r;
    if (is_identity()) {
      r.x = F::zero();
      r.y = F::zero();
      return r;
    }
    // one inversion: (zz*zzz)^-1, then zz^-1 = inv*zzz, zzz^-1 = inv*zz
    F i = (zz * zzz).inv();
    r.x = x * (i * zzz);
    r.y = y * (i * zz);
    return r;
  }
};

}  // namespace nmx
