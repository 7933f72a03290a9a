// host_fp4.hpp -- HOST-ONLY field arithmetic on 4 x 64-bit limbs (Montgomery, R = 2^256) for the O(1)-per-round algebra and
// the tail rounds of the sum-check provers (sumcheck_prove.hpp).
//
// The device form (fp.hpp: 9 x 29-bit limbs, R' = 2^261) is built for v_mad_u64_u32 chains; its portable host build costs
// ~250 ns per product.  A sum-check round needs ~40 host products between two kernel launches (derive_from_claim_deg2/1,
// UniPoly::from_evals_deg3/2, evaluate, bound: /root/reference/src/spartan/sumcheck.rs:680-753,1226-1231,
// polys/univariate.rs:90-149) -- 10 us of a 25 us round.  Here a product is one 4 x 4 schoolbook + Montgomery reduction on
// unsigned __int128 (~25 ns), which is also what lets the last rounds of a proof (tables of <= 64 elements) finish on the
// host in less time than ONE kernel round trip.  The moduli come from FpParams (fp.hpp); everything else is derived here.
#pragma once
#include <stdint.h>
#include <string.h>

#include "fp.hpp"

namespace nmx {

template <int FID> struct HostFp4 {
  using PP = FpParams<FID>;
  typedef unsigned __int128 u128;
  uint64_t v[4];  // value * 2^256 mod p, < p

  struct Consts {
    uint64_t p[4], ninv, one[4], r2[4];  // p, -p^-1 mod 2^64, 2^256 mod p, 2^512 mod p
  };
  static const Consts& C() {
    static const Consts c = [] {
      Consts k;
      for (int i = 0; i < 4; i++) k.p[i] = (uint64_t)PP::PW[2 * i] | ((uint64_t)PP::PW[2 * i + 1] << 32);
      uint64_t x = 1;  // Newton: x = p^-1 mod 2^64
      for (int i = 0; i < 6; i++) x *= 2 - k.p[0] * x;
      k.ninv = (uint64_t)0 - x;
      uint64_t t[4] = {1, 0, 0, 0};
      for (int i = 0; i < 512; i++) {
        dbl_raw(t, k.p);
        if (i == 255) memcpy(k.one, t, 32);
      }
      memcpy(k.r2, t, 32);
      return k;
    }();
    return c;
  }
  // t = 2 t mod p for t < p (raw integers)
  static void dbl_raw(uint64_t t[4], const uint64_t p[4]) {
    const uint64_t top = t[3] >> 63;
    for (int i = 3; i > 0; i--) t[i] = (t[i] << 1) | (t[i - 1] >> 63);
    t[0] <<= 1;
    if (top || geq(t, p)) sub_raw(t, p);
  }
  static bool geq(const uint64_t a[4], const uint64_t b[4]) {
    for (int i = 3; i >= 0; i--)
      if (a[i] != b[i]) return a[i] > b[i];
    return true;
  }
  static void sub_raw(uint64_t a[4], const uint64_t b[4]) {
    u128 bw = 0;
    for (int i = 0; i < 4; i++) {
      const u128 d = (u128)a[i] - b[i] - bw;
      a[i] = (uint64_t)d;
      bw = (d >> 64) & 1;
    }
  }

  static HostFp4 zero() {
    HostFp4 r;
    memset(r.v, 0, 32);
    return r;
  }
  static HostFp4 one() {
    HostFp4 r;
    memcpy(r.v, C().one, 32);
    return r;
  }
  bool is_zero() const { return (v[0] | v[1] | v[2] | v[3]) == 0; }
  friend bool operator==(const HostFp4& a, const HostFp4& b) { return memcmp(a.v, b.v, 32) == 0; }

  friend HostFp4 operator+(const HostFp4& a, const HostFp4& b) {
    HostFp4 r;
    u128 c = 0;
    for (int i = 0; i < 4; i++) {
      c += (u128)a.v[i] + b.v[i];
      r.v[i] = (uint64_t)c;
      c >>= 64;
    }
    if (c || geq(r.v, C().p)) sub_raw(r.v, C().p);
    return r;
  }
  friend HostFp4 operator-(const HostFp4& a, const HostFp4& b) {
    HostFp4 r;
    u128 bw = 0;
    for (int i = 0; i < 4; i++) {
      const u128 d = (u128)a.v[i] - b.v[i] - bw;
      r.v[i] = (uint64_t)d;
      bw = (d >> 64) & 1;
    }
    if (bw) {
      u128 c = 0;
      for (int i = 0; i < 4; i++) {
        c += (u128)r.v[i] + C().p[i];
        r.v[i] = (uint64_t)c;
        c >>= 64;
      }
    }
    return r;
  }
  HostFp4 dbl() const { return *this + *this; }
  // Montgomery product a b / 2^256 mod p (CIOS, 4 limbs)
  static void mont_mul(uint64_t out[4], const uint64_t a[4], const uint64_t b[4]) {
    const Consts& k = C();
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
      u128 c = 0;
      for (int j = 0; j < 4; j++) {
        c += (u128)a[j] * b[i] + t[j];
        t[j] = (uint64_t)c;
        c >>= 64;
      }
      c += t[4];
      t[4] = (uint64_t)c;
      t[5] = (uint64_t)(c >> 64);
      const uint64_t m = t[0] * k.ninv;
      c = (u128)m * k.p[0] + t[0];
      c >>= 64;
      for (int j = 1; j < 4; j++) {
        c += (u128)m * k.p[j] + t[j];
        t[j - 1] = (uint64_t)c;
        c >>= 64;
      }
      c += t[4];
      t[3] = (uint64_t)c;
      t[4] = t[5] + (uint64_t)(c >> 64);
    }
    memcpy(out, t, 32);
    if (t[4] || geq(out, k.p)) sub_raw(out, k.p);
  }
  friend HostFp4 operator*(const HostFp4& a, const HostFp4& b) {
    HostFp4 r;
    mont_mul(r.v, a.v, b.v);
    return r;
  }
  // --- words <-> value
  static void load(uint64_t t[4], const void* words32) { memcpy(t, words32, 32); }
  // a plain integer < p (canonical bytes) -> element
  static HostFp4 from_canonical(const void* words32) {
    uint64_t t[4];
    load(t, words32);
    HostFp4 r;
    mont_mul(r.v, t, C().r2);
    return r;
  }
  // halo2curves' in-memory form (x 2^256 mod p) IS this form
  static HostFp4 from_mont256(const void* words32) {
    HostFp4 r;
    load(r.v, words32);
    return r;
  }
  void to_canonical(void* words32) const {
    const uint64_t o[4] = {1, 0, 0, 0};
    uint64_t t[4];
    mont_mul(t, v, o);
    memcpy(words32, t, 32);
  }
  void to_mont256(void* words32) const { memcpy(words32, v, 32); }
  // plain integer X and an element k -> the ELEMENT X * value(k) / 2^256 (one Montgomery product of the raw words with k's
  // residue: no conversion of X first).  The device's raw sums come in through this with k = 2^e (sumcheck_prove.hpp `raw`).
  static HostFp4 from_plain_times(const void* words32, const HostFp4& k) {
    uint64_t t[4];
    load(t, words32);
    if (geq(t, C().p)) sub_raw(t, C().p);  // device sums are canonical; be safe
    HostFp4 r;
    mont_mul(r.v, t, k.v);
    return r;
  }
  // the element 2^e (e >= 0)
  static HostFp4 pow2(uint32_t e) {
    HostFp4 r = one();
    for (uint32_t i = 0; i < e; i++) dbl_raw(r.v, C().p);
    return r;
  }
  static HostFp4 from_u64(uint64_t x) {
    const uint64_t t[4] = {x, 0, 0, 0};
    HostFp4 r;
    mont_mul(r.v, t, C().r2);
    return r;
  }
  // the device's internal residue (value * 2^261 mod p as 9 x 29-bit limbs): five doublings of value * 2^256
  Fp<FID> to_device() const {
    uint64_t t[4];
    memcpy(t, v, 32);
    for (int i = 0; i < 5; i++) dbl_raw(t, C().p);
    uint32_t w[8];
    memcpy(w, t, 32);
    return Fp<FID>::from_words(w);
  }
  // ---- inversion ------------------------------------------------------------------------------------------------------
  // y^-1 mod p for 0 < y < p as plain integers: the binary extended GCD with 31 steps at a time decided on 64-bit approximations
  // of (a, b) -- their top 33 and low 31 bits -- and applied to the full values as ONE linear combination with 32-bit
  // coefficients (T. Pornin, "Optimized Binary GCD for Modular Inversion", 2020; variable time here: nothing secret goes through
  // the provers' host algebra).  ~17 outer steps for a 254-bit modulus: 1-2 us against 5-9 for the bit-at-a-time form
  // (fp.hpp inv_words_host), which a sum-check round spends once (sc_host.hpp Eq::prepare).  false: did not converge (the
  // caller falls back).
  typedef __int128 i128;
  // out (5 limbs, two's complement) = f * A + g * B for |f|, |g| <= 2^31 and unsigned 4-limb A, B
  static void lincomb5(int64_t f, const uint64_t A[4], int64_t g, const uint64_t B[4], uint64_t out[5]) {
    const uint64_t fa = (uint64_t)(f < 0 ? -f : f), ga = (uint64_t)(g < 0 ? -g : g);
    uint64_t P[5], Q[5];
    u128 c = 0;
    for (int i = 0; i < 4; i++) {
      c += (u128)A[i] * fa;
      P[i] = (uint64_t)c;
      c >>= 64;
    }
    P[4] = (uint64_t)c;
    c = 0;
    for (int i = 0; i < 4; i++) {
      c += (u128)B[i] * ga;
      Q[i] = (uint64_t)c;
      c >>= 64;
    }
    Q[4] = (uint64_t)c;
    auto neg5 = [](uint64_t x[5]) {
      u128 k = 1;
      for (int i = 0; i < 5; i++) {
        k += (u128)(~x[i]);
        x[i] = (uint64_t)k;
        k >>= 64;
      }
    };
    if (f < 0) neg5(P);
    if (g < 0) neg5(Q);
    c = 0;
    for (int i = 0; i < 5; i++) {
      c += (u128)P[i] + Q[i];
      out[i] = (uint64_t)c;
      c >>= 64;
    }
  }
  // x (5 limbs, two's complement, a multiple of 2^31) >> 31 into 4 limbs + returns the sign (true: negative; the 4 limbs then hold
  // the low 256 bits of the two's complement quotient)
  static bool sar31(const uint64_t x[5], uint64_t out[4]) {
    for (int i = 0; i < 4; i++) out[i] = (x[i] >> 31) | (x[i + 1] << 33);
    return (int64_t)x[4] < 0;
  }
  static void neg4(uint64_t x[4]) {
    u128 k = 1;
    for (int i = 0; i < 4; i++) {
      k += (u128)(~x[i]);
      x[i] = (uint64_t)k;
      k >>= 64;
    }
  }
  static int bitlen4(const uint64_t x[4]) {
    for (int i = 3; i >= 0; i--)
      if (x[i]) return 64 * i + 64 - __builtin_clzll(x[i]);
    return 0;
  }
  // bits [lo, lo + 33) of x (lo >= 0)
  static uint64_t bits33(const uint64_t x[4], int lo) {
    const int w = lo >> 6, o = lo & 63;
    uint64_t v = x[w] >> o;
    if (o > 31 && w < 3) v |= x[w + 1] << (64 - o);
    return v & 0x1ffffffffull;
  }
  static bool inv_plain(const uint64_t y[4], uint64_t out[4]) {
    const Consts& k = C();
    uint64_t a[4], b[4], u[4] = {1, 0, 0, 0}, v[4] = {0, 0, 0, 0};
    memcpy(a, y, 32), memcpy(b, k.p, 32);
    const uint64_t m31 = (uint64_t)0 - k.ninv;  // p^-1 mod 2^64 (its low 31 bits are what is used)
    for (int outer = 0; outer < 40; outer++) {
      if ((a[0] | a[1] | a[2] | a[3]) == 0) {
        if (!(b[0] == 1 && (b[1] | b[2] | b[3]) == 0)) return false;  // gcd != 1 (cannot happen for 0 < y < p, p prime)
        memcpy(out, v, 32);
        return true;
      }
      const int la = bitlen4(a), lb = bitlen4(b), n = la > lb ? la : lb;
      uint64_t xa, xb;
      if (n <= 64) {
        xa = a[0], xb = b[0];
      } else {
        xa = (bits33(a, n - 33) << 31) | (a[0] & 0x7fffffffu);
        xb = (bits33(b, n - 33) << 31) | (b[0] & 0x7fffffffu);
      }
      int64_t f0 = 1, g0 = 0, f1 = 0, g1 = 1;
      for (int i = 0; i < 31; i++) {  // masks instead of branches: the two tests are coin flips to a predictor
        const uint64_t odd = (uint64_t)0 - (xa & 1), sw = odd & ((uint64_t)0 - (uint64_t)(xa < xb));
        const uint64_t tx = (xa ^ xb) & sw, tf = (uint64_t)(f0 ^ f1) & sw, tg = (uint64_t)(g0 ^ g1) & sw;
        xa ^= tx, xb ^= tx;
        f0 = (int64_t)((uint64_t)f0 ^ tf), f1 = (int64_t)((uint64_t)f1 ^ tf);
        g0 = (int64_t)((uint64_t)g0 ^ tg), g1 = (int64_t)((uint64_t)g1 ^ tg);
        xa -= xb & odd;
        f0 -= (int64_t)((uint64_t)f1 & odd), g0 -= (int64_t)((uint64_t)g1 & odd);
        xa >>= 1;
        f1 = (int64_t)((uint64_t)f1 << 1), g1 = (int64_t)((uint64_t)g1 << 1);
      }
      // (a, b) <- (f0 a + g0 b, f1 a + g1 b) / 2^31, made non-negative
      uint64_t t5[5], na[4], nb[4];
      lincomb5(f0, a, g0, b, t5);
      if (sar31(t5, na)) neg4(na), f0 = -f0, g0 = -g0;
      lincomb5(f1, a, g1, b, t5);
      if (sar31(t5, nb)) neg4(nb), f1 = -f1, g1 = -g1;
      // (u, v) <- the same combinations / 2^31 mod p: a multiple of p clears the low 31 bits first
      auto comb_mod = [&](int64_t f, int64_t g, uint64_t res[4]) {
        uint64_t t[5];
        lincomb5(f, u, g, v, t);
        const uint64_t q = ((uint64_t)0 - t[0] * m31) & 0x7fffffffu;  // t + q p == 0 mod 2^31
        u128 c = 0;
        for (int i = 0; i < 4; i++) {
          c += (u128)k.p[i] * q + t[i];
          t[i] = (uint64_t)c;
          c >>= 64;
        }
        t[4] += (uint64_t)c;  // two's complement: the carry simply adds in
        const bool negative = sar31(t, res);  // in (-p, 2p)
        if (negative) {
          u128 cc = 0;
          for (int i = 0; i < 4; i++) {
            cc += (u128)res[i] + k.p[i];
            res[i] = (uint64_t)cc;
            cc >>= 64;
          }
        } else if (geq(res, k.p)) {
          sub_raw(res, k.p);
        }
      };
      uint64_t nu[4], nv[4];
      comb_mod(f0, g0, nu);
      comb_mod(f1, g1, nv);
      memcpy(a, na, 32), memcpy(b, nb, 32), memcpy(u, nu, 32), memcpy(v, nv, 32);
    }
    return false;
  }
  // 1 / x (0 -> 0).  The fast inversion's answer is checked by one product; anything unexpected goes to the bit-at-a-time
  // extended Euclid (fp.hpp inv_words_host).
  HostFp4 inv() const {
    if (is_zero()) return zero();
#if defined(__HIP_DEVICE_COMPILE__)
    return zero();  // host-only type: this body only exists for the device pass of hipcc to parse
#else
    {
      uint64_t z4[4];
      if (inv_plain(v, z4)) {  // (x 2^256)^-1 as a plain integer
        HostFp4 r;
        mont_mul(r.v, z4, C().r2);   // x^-1 2^-256 2^512 / 2^256 = x^-1 (plain)
        mont_mul(r.v, r.v, C().r2);  // x^-1 2^256
        if (r * *this == one()) return r;
      }
    }
    return inv_slow();
#endif
  }
  HostFp4 inv_slow() const {
    if (is_zero()) return zero();
#if defined(__HIP_DEVICE_COMPILE__)
    return zero();
#else
    uint32_t w[8], z[8];
    memcpy(w, v, 32);
    Fp<FID>::inv_words_host(w, z);  // (x 2^256)^-1
    uint64_t t[4];
    memcpy(t, z, 32);
    HostFp4 r;
    mont_mul(r.v, t, C().r2);    // x^-1 2^-256 2^512 / 2^256 = x^-1
    mont_mul(r.v, r.v, C().r2);  // x^-1 2^256
    return r;
#endif
  }
};

}  // namespace nmx
