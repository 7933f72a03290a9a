// curves.hpp -- the four curves of the hot path and byte marshalling at the C-ABI boundary.
//   BN254 G1 / Grumpkin : /root/reference/src/provider/bn256_grumpkin.rs:26-33,35-41,80-86
//   Pallas / Vesta      : /root/reference/src/provider/pasta.rs:24-47
// Curve constant b never enters the a = 0 addition formulas (only the NMX_BASES_VALIDATE on-curve check uses it);
// generators are only used by nmx_bases_generate.
#pragma once
#include <string.h>
#include "curve.hpp"

namespace nmx {

template <int CID> struct CurveT;
// canonical little-endian limbs of the generator (x, y)
template <> struct CurveT<0> {  // BN254 G1: y^2 = x^3 + 3, G = (1, 2)
  static constexpr int BF = F_BN254_FQ, SF = F_BN254_FR;
  static constexpr uint32_t GX[8] = {1, 0, 0, 0, 0, 0, 0, 0};
  static constexpr uint32_t GY[8] = {2, 0, 0, 0, 0, 0, 0, 0};
  static constexpr uint32_t B[8] = {3, 0, 0, 0, 0, 0, 0, 0};
};
template <> struct CurveT<1> {  // Grumpkin: y^2 = x^3 - 17 over BN254 Fr, G = (1, sqrt(-16))
  static constexpr int BF = F_BN254_FR, SF = F_BN254_FQ;
  static constexpr uint32_t GX[8] = {1, 0, 0, 0, 0, 0, 0, 0};
  // 0x2cf135e7506a45d632d270d45f1181294833fc48d823f272c
  static constexpr uint32_t GY[8] = {0x823f272cu, 0x833fc48du, 0xf1181294u, 0x2d270d45u,
                                     0x06a45d63u, 0xcf135e75u, 0x00000002u, 0x00000000u};
  // -17 mod r
  static constexpr uint32_t B[8] = {0xeffffff0u, 0x43e1f593u, 0x79b97091u, 0x2833e848u,
                                    0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
};
template <> struct CurveT<2> {  // Pallas: y^2 = x^3 + 5, G = (-1, 2)
  static constexpr int BF = F_PASTA_FP, SF = F_PASTA_FQ;
  static constexpr uint32_t GX[8] = {0x00000000u, 0x992d30edu, 0x094cf91bu, 0x224698fcu,
                                     0x00000000u, 0x00000000u, 0x00000000u, 0x40000000u};
  static constexpr uint32_t GY[8] = {2, 0, 0, 0, 0, 0, 0, 0};
  static constexpr uint32_t B[8] = {5, 0, 0, 0, 0, 0, 0, 0};
};
template <> struct CurveT<3> {  // Vesta: y^2 = x^3 + 5, G = (-1, 2)
  static constexpr int BF = F_PASTA_FQ, SF = F_PASTA_FP;
  static constexpr uint32_t GX[8] = {0x00000000u, 0x8c46eb21u, 0x0994a8ddu, 0x224698fcu,
                                     0x00000000u, 0x00000000u, 0x00000000u, 0x40000000u};
  static constexpr uint32_t GY[8] = {2, 0, 0, 0, 0, 0, 0, 0};
  static constexpr uint32_t B[8] = {5, 0, 0, 0, 0, 0, 0, 0};
};

// ---- host-side byte marshalling (x86-64 is little-endian: 8 x u32 words map 1:1 onto 32 LE bytes) ---------
// 32 bytes -> unpacked limbs, no reduction (the value is whatever integer the bytes spell)
template <int FID> inline Fp<FID> fp_from_bytes(const uint8_t* b) {
  uint32_t w[8];
  memcpy(w, b, 32);
  return Fp<FID>::from_words(w);
}
// normalized value < 2^256 -> 32 bytes
template <int FID> inline void fp_to_bytes(const Fp<FID>& f, uint8_t* b) {
  uint32_t w[8];
  f.to_words(w);
  memcpy(b, w, 32);
}

// XYZZ (internal form) -> canonical affine x||y + inf flag: `to_coordinates()` (traits.rs:303-312)
template <int FID> inline void xyzz_to_xy64(const XYZZ<FID>& p, uint8_t* out, uint8_t* is_inf) {
  if (p.is_identity()) {
    memset(out, 0, 64);
    if (is_inf) *is_inf = 1;
    return;
  }
  Affine<FID> a = p.to_affine();
  fp_to_bytes(a.x.to_canonical(), out);
  fp_to_bytes(a.y.to_canonical(), out + 32);
  if (is_inf) *is_inf = 0;
}

// k * P for a canonical 256-bit integer k given as 8 x u32 (host tail only: the h * r term of commit)
template <int FID> inline XYZZ<FID> scalar_mul(const XYZZ<FID>& p, const uint32_t k[8]) {
  XYZZ<FID> acc = XYZZ<FID>::identity();
  for (int i = 255; i >= 0; i--) {
    acc.dbl_in_place();
    if ((k[i >> 5] >> (i & 31)) & 1u) acc.add(p);
  }
  return acc;
}

}  // namespace nmx
