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
  // 1 / x (0 -> 0): binary extended Euclid on the residue (fp.hpp inv_words_host), then back to Montgomery form
  HostFp4 inv() const {
    if (is_zero()) return zero();
#if defined(__HIP_DEVICE_COMPILE__)
    return zero();  // host-only type: this body only exists for the device pass of hipcc to parse
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
