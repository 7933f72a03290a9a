// fp.hpp -- 256-bit prime-field arithmetic in Montgomery form (R = 2^256) on 8 x 32-bit limbs.
//
// One element per GPU lane, all 8 limbs in VGPRs.  The in-memory layout (8 little-endian u32 = 4
// little-endian u64 = 32 bytes) is bit-identical to halo2curves' Montgomery representation, which is what
// the reference keeps in its `Vec<Scalar>` / `Vec<G1Affine>` ("4 x u64 Montgomery form,
// R = 2^256"), so the "raw Montgomery" ABI flag is zero-copy.
//
// Multiplication is product-scanning Montgomery (FIPS) with a 96-bit column accumulator: every partial
// product is one v_mad_u64_u32 (32x32+64 -> 64, carry-out to an SGPR pair) plus one v_addc for the top word.
// 136 multiply-adds per modmul; no MFMA, no floating point (north_star: integer modular arithmetic).
//
// The same source compiles for the host (g++ / hipcc host pass): the host tail of an MSM (Horner combine
// of <= 32 window sums, one inversion) and the CPU-side unit tests of this header use it unchanged.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIP__)
#define NMX_HD __host__ __device__ __forceinline__
#else
#define NMX_HD inline __attribute__((always_inline))
#endif

namespace nmx {

// Field ids.  Moduli are the hex strings of /root/reference/src/provider/bn256_grumpkin.rs:39-40 and
// /root/reference/src/provider/pasta.rs:37-38.
enum FieldId : int { F_BN254_FQ = 0, F_BN254_FR = 1, F_PASTA_FP = 2, F_PASTA_FQ = 3 };

template <int FID> struct FpParams;

template <> struct FpParams<F_BN254_FQ> {
  static constexpr int BITS = 254;
  static constexpr uint32_t NINV = 0xe4866389u;  // -p^-1 mod 2^32
  static constexpr uint32_t P[8] = {0xd87cfd47u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u,
                                    0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
  static constexpr uint32_t R1[8] = {0xc58f0d9du, 0xd35d438du, 0xf5c70b3du, 0x0a78eb28u,
                                     0x7879462cu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
  static constexpr uint32_t R2[8] = {0x538afa89u, 0xf32cfc5bu, 0xd44501fbu, 0xb5e71911u,
                                     0x0a417ff6u, 0x47ab1effu, 0xcab8351fu, 0x06d89f71u};
};
template <> struct FpParams<F_BN254_FR> {
  static constexpr int BITS = 254;
  static constexpr uint32_t NINV = 0xefffffffu;
  static constexpr uint32_t P[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u,
                                    0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
  static constexpr uint32_t R1[8] = {0x4ffffffbu, 0xac96341cu, 0x9f60cd29u, 0x36fc7695u,
                                     0x7879462eu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
  static constexpr uint32_t R2[8] = {0xae216da7u, 0x1bb8e645u, 0xe35c59e3u, 0x53fe3ab1u,
                                     0x53bb8085u, 0x8c49833du, 0x7f4e44a5u, 0x0216d0b1u};
};
template <> struct FpParams<F_PASTA_FP> {
  static constexpr int BITS = 255;
  static constexpr uint32_t NINV = 0xffffffffu;
  static constexpr uint32_t P[8] = {0x00000001u, 0x992d30edu, 0x094cf91bu, 0x224698fcu,
                                    0x00000000u, 0x00000000u, 0x00000000u, 0x40000000u};
  static constexpr uint32_t R1[8] = {0xfffffffdu, 0x34786d38u, 0xe41914adu, 0x992c350bu,
                                     0xffffffffu, 0xffffffffu, 0xffffffffu, 0x3fffffffu};
  static constexpr uint32_t R2[8] = {0x0000000fu, 0x8c78ecb3u, 0x8b0de0e7u, 0xd7d30dbdu,
                                     0xc3c95d18u, 0x7797a99bu, 0x7b9cb714u, 0x096d41afu};
};
template <> struct FpParams<F_PASTA_FQ> {
  static constexpr int BITS = 255;
  static constexpr uint32_t NINV = 0xffffffffu;
  static constexpr uint32_t P[8] = {0x00000001u, 0x8c46eb21u, 0x0994a8ddu, 0x224698fcu,
                                    0x00000000u, 0x00000000u, 0x00000000u, 0x40000000u};
  static constexpr uint32_t R1[8] = {0xfffffffdu, 0x5b2b3e9cu, 0xe3420567u, 0x992c350bu,
                                     0xffffffffu, 0xffffffffu, 0xffffffffu, 0x3fffffffu};
  static constexpr uint32_t R2[8] = {0x0000000fu, 0xfc9678ffu, 0x891a16e3u, 0x67bb433du,
                                     0x04ccf590u, 0x7fae2310u, 0x7ccfdaa9u, 0x096d41afu};
};

// 96-bit column accumulator: acc += x*y
struct Acc96 {
  uint64_t lo;
  uint32_t hi;
};
NMX_HD void mac(Acc96& a, uint32_t x, uint32_t y) {
  uint64_t prod_sum;
  // (u64)x*y + a.lo, carry-out into a.hi : v_mad_u64_u32 + v_addc_co_u32 on gfx950
  bool c = __builtin_add_overflow((uint64_t)x * (uint64_t)y, a.lo, &prod_sum);
  a.lo = prod_sum;
  a.hi += (uint32_t)c;
}
NMX_HD void acc_shift(Acc96& a) {
  a.lo = (a.lo >> 32) | ((uint64_t)a.hi << 32);
  a.hi = 0;
}

template <int FID> struct Fp {
  using PP = FpParams<FID>;
  uint32_t l[8];

  static NMX_HD Fp zero() {
    Fp r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = 0;
    return r;
  }
  static NMX_HD Fp one() {  // Montgomery form of 1
    Fp r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = PP::R1[i];
    return r;
  }
  static NMX_HD Fp r2() {
    Fp r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = PP::R2[i];
    return r;
  }
  NMX_HD bool is_zero() const {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) o |= l[i];
    return o == 0;
  }
  NMX_HD bool operator==(const Fp& b) const {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) o |= (l[i] ^ b.l[i]);
    return o == 0;
  }
  NMX_HD bool operator!=(const Fp& b) const { return !(*this == b); }

  // r = a - P if a >= P (a < 2P)
  NMX_HD void cond_sub_p() {
    uint32_t t[8];
    uint64_t bw = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      uint64_t d = (uint64_t)l[i] - PP::P[i] - bw;
      t[i] = (uint32_t)d;
      bw = (d >> 32) & 1u;
    }
    if (bw == 0) {
#pragma unroll
      for (int i = 0; i < 8; i++) l[i] = t[i];
    }
  }

  friend NMX_HD Fp operator+(const Fp& a, const Fp& b) {
    Fp r;
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      c += (uint64_t)a.l[i] + b.l[i];
      r.l[i] = (uint32_t)c;
      c >>= 32;
    }
    // a, b < P < 2^255  =>  a + b < 2^256: no carry out of limb 7
    r.cond_sub_p();
    return r;
  }
  friend NMX_HD Fp operator-(const Fp& a, const Fp& b) {
    Fp r;
    uint64_t bw = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      uint64_t d = (uint64_t)a.l[i] - b.l[i] - bw;
      r.l[i] = (uint32_t)d;
      bw = (d >> 32) & 1u;
    }
    if (bw) {
      uint64_t c = 0;
#pragma unroll
      for (int i = 0; i < 8; i++) {
        c += (uint64_t)r.l[i] + PP::P[i];
        r.l[i] = (uint32_t)c;
        c >>= 32;
      }
    }
    return r;
  }
  NMX_HD Fp neg() const { return is_zero() ? *this : (zero() - *this); }
  NMX_HD Fp dbl() const { return *this + *this; }

  // Montgomery product a*b*R^-1 mod P, fully reduced.
  friend NMX_HD Fp operator*(const Fp& a, const Fp& b) {
    Acc96 acc{0, 0};
    uint32_t m[8];
    Fp r;
#pragma unroll
    for (int k = 0; k < 8; k++) {
#pragma unroll
      for (int i = 0; i <= k; i++) mac(acc, a.l[i], b.l[k - i]);
#pragma unroll
      for (int i = 0; i < k; i++) mac(acc, m[i], PP::P[k - i]);
      m[k] = (uint32_t)acc.lo * PP::NINV;
      mac(acc, m[k], PP::P[0]);
      acc_shift(acc);
    }
#pragma unroll
    for (int k = 8; k < 16; k++) {
#pragma unroll
      for (int i = k - 7; i < 8; i++) mac(acc, a.l[i], b.l[k - i]);
#pragma unroll
      for (int i = k - 7; i < 8; i++) mac(acc, m[i], PP::P[k - i]);
      r.l[k - 8] = (uint32_t)acc.lo;
      acc_shift(acc);
    }
    // a,b < P and P < 2^255  =>  (ab + mP)/R < 2P < 2^256: the 9th word is zero
    r.cond_sub_p();
    return r;
  }
  NMX_HD Fp sqr() const { return (*this) * (*this); }

  // canonical (non-Montgomery) integer -> Montgomery form, input must be < P
  NMX_HD Fp to_mont() const { return (*this) * r2(); }
  // Montgomery form -> canonical integer
  NMX_HD Fp from_mont() const {
    Fp o = zero();
    o.l[0] = 1;
    return (*this) * o;
  }
  NMX_HD bool lt_p() const {  // canonical-range check
    uint64_t bw = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      uint64_t d = (uint64_t)l[i] - PP::P[i] - bw;
      bw = (d >> 32) & 1u;
    }
    return bw != 0;
  }

  // x^(P-2): Fermat inversion; 0 -> 0.  Used O(1) times per MSM (host tail) and in batch normalisation.
  NMX_HD Fp inv() const {
    uint32_t e[8];
    uint64_t bw = 2;
    for (int i = 0; i < 8; i++) {
      uint64_t d = (uint64_t)PP::P[i] - bw;
      e[i] = (uint32_t)d;
      bw = (d >> 32) & 1u;
    }
    Fp acc = one();
    for (int i = 255; i >= 0; i--) {
      acc = acc.sqr();
      if ((e[i >> 5] >> (i & 31)) & 1u) acc = acc * (*this);
    }
    return acc;
  }
};

}  // namespace nmx
