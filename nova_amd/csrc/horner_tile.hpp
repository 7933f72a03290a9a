// horner_tile.hpp -- suffix Horner  out[i] = f[i] + u * out[i + 1]  (poly_eval / div_by_monomial of HyperKZG,
// /root/reference/src/provider/hyperkzg.rs:946-1020) as two coalesced passes over tiles.  Included by fieldvec.hip only.
//
// Round 1 gave every lane a 16-coefficient chunk and walked it straight from HBM: consecutive lanes read addresses 512 B
// apart (every 32-byte access its own cache line), the local values were written, read back and fixed up, and the
// carries recursed over five or six levels of launches: 0.90 ms for 2^24 coefficients = 15 % of the HBM roofline,
// 0.12 ms for 2^20.  Here a block of 256 lanes owns a TILE of 2048 consecutive coefficients:
//   k_horner_heads   tile -> LDS with coalesced 16-byte loads (transposed so that lane l then reads ITS eight
//                    coefficients conflict-free), lane-local Horner -> lane head H_l (stored: 4 B per coefficient),
//                    LDS tree over the 256 heads with the powers u^8, u^16, ... -> tile head
//   (recursion)      suffix Horner of the tile heads with u^2048 -> carry into every tile (2^24 coefficients: 8192
//                    heads, then 4: three levels instead of six)
//   k_horner_apply   tile -> LDS again, suffix scan of the lane heads across the block (Hillis-Steele, eight steps,
//                    the tile's carry entering at lane 255), every lane re-walks its eight coefficients from its carry
//                    and leaves the results in LDS, which goes out with coalesced 16-byte stores
// 104 B of traffic per coefficient (f twice, out once, lane heads once each way) and 3.2 field multiplications
// (8 + 1 in the first pass, 1 + 8 + 8 in the second, per eight coefficients).
#pragma once
#include "runtime.hpp"

namespace nmx {

static constexpr uint32_t kHtThreads = 256, kHtPer = 8, kHtTile = kHtThreads * kHtPer;  // 2048 coefficients per tile
struct alignas(16) HtPiece {
  uint32_t w[4];
};
// powers of one recursion level, internal form, canonical: [0] = u, [1 + k] = u^(8 * 2^k), k = 0..7
struct HtPowers {
  uint32_t w[9][8];
};

template <int FID> struct HtShared {
  HtPiece stage[2 * kHtPer][kHtThreads + 1];  // row = half * 8 + step, column = lane (+1: rows land on different banks)
  uint32_t xch[9][kHtThreads];                // limb-major exchange buffer for the cross-lane steps
};

template <int FID> __device__ __forceinline__ void ht_load_tile(HtShared<FID>& sh, const uint32_t* f, uint32_t n, uint32_t tile) {
  const HtPiece* src = reinterpret_cast<const HtPiece*>(f);
  const uint32_t t = threadIdx.x;
  const size_t base = (size_t)tile * kHtTile;
#pragma unroll
  for (uint32_t k = 0; k < 2 * kHtPer; k++) {
    const uint32_t g = t + kHtThreads * k, e = g >> 1, q = g & 1u, l = e / kHtPer, s = e % kHtPer;
    HtPiece v{{0, 0, 0, 0}};
    if (base + e < n) v = src[2 * base + g];
    sh.stage[q * kHtPer + s][l] = v;
  }
  __syncthreads();
}
template <int FID> __device__ __forceinline__ Fp<FID> ht_elem(const HtShared<FID>& sh, uint32_t s, uint32_t l) {
  uint32_t w[8];
  const HtPiece a = sh.stage[s][l], b = sh.stage[kHtPer + s][l];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    w[i] = a.w[i];
    w[4 + i] = b.w[i];
  }
  return Fp<FID>::from_words(w);
}
template <int FID> __device__ __forceinline__ void ht_put(HtShared<FID>& sh, uint32_t* xch_base, uint32_t l, const Fp<FID>& v) {
#pragma unroll
  for (int i = 0; i < 9; i++) xch_base[i * kHtThreads + l] = v.l[i];
}
template <int FID> __device__ __forceinline__ Fp<FID> ht_get(const uint32_t* xch_base, uint32_t l) {
  Fp<FID> v;
#pragma unroll
  for (int i = 0; i < 9; i++) v.l[i] = xch_base[i * kHtThreads + l];
  return v;
}

template <int FID>
__global__ __launch_bounds__(256) void k_horner_heads(const uint32_t* f, uint32_t n, HtPowers pw, uint32_t* lane_heads,
                                                      uint32_t* tile_heads) {
  using F = Fp<FID>;
  __shared__ HtShared<FID> sh;
  const uint32_t l = threadIdx.x, tile = blockIdx.x;
  ht_load_tile<FID>(sh, f, n, tile);
  const F u = F::from_words(pw.w[0]);
  F h = F::zero();
#pragma unroll
  for (uint32_t s = kHtPer; s-- > 0;) h = (ht_elem<FID>(sh, s, l) + u * h).norm();  // < 2.01 p
  h = h.canon();
  h.to_words(lane_heads + 8 * ((size_t)tile * kHtThreads + l));
  // tile head = sum_l (u^8)^l * H_l: pairwise tree, step k combines lanes 2^k apart with u^(8 * 2^k)
  uint32_t* x = &sh.xch[0][0];
  for (uint32_t k = 0; k < 8; k++) {
    const uint32_t d = 1u << k;
    ht_put<FID>(sh, x, l, h);
    __syncthreads();
    if ((l & (2 * d - 1)) == 0) h = (h + F::from_words(pw.w[1 + k]) * ht_get<FID>(x, l + d)).norm().canon();
    __syncthreads();
  }
  if (l == 0) h.to_words(tile_heads + 8 * (size_t)tile);
}

template <int FID>
__global__ __launch_bounds__(256) void k_horner_apply(const uint32_t* f, uint32_t n, HtPowers pw, const uint32_t* lane_heads,
                                                      const uint32_t* carries /* null: one tile */, uint32_t ntiles,
                                                      uint32_t* out) {
  using F = Fp<FID>;
  __shared__ HtShared<FID> sh;
  const uint32_t l = threadIdx.x, tile = blockIdx.x;
  ht_load_tile<FID>(sh, f, n, tile);
  const F u = F::from_words(pw.w[0]);
  // X = the suffix value at the first coefficient of the next tile
  F X = F::zero();
  if (carries && tile + 1 < ntiles) X = F::from_words(carries + 8 * ((size_t)tile + 1));
  F t = F::from_words(lane_heads + 8 * ((size_t)tile * kHtThreads + l));
  if (l == kHtThreads - 1) t = (t + F::from_words(pw.w[1]) * X).norm().canon();  // the carry enters behind the last lane
  // suffix scan over the lanes: T_l = H_l + u^8 * T_{l+1}
  uint32_t* x = &sh.xch[0][0];
  for (uint32_t k = 0; k < 8; k++) {
    const uint32_t d = 1u << k;
    ht_put<FID>(sh, x, l, t);
    __syncthreads();
    if (l + d < kHtThreads) t = (t + F::from_words(pw.w[1 + k]) * ht_get<FID>(x, l + d)).norm().canon();
    __syncthreads();
  }
  // carry into this lane's chunk = T_{l+1} (the tile's carry for the last lane)
  ht_put<FID>(sh, x, l, t);
  __syncthreads();
  F c = l + 1 < kHtThreads ? ht_get<FID>(x, l + 1) : X;
#pragma unroll
  for (uint32_t s = kHtPer; s-- > 0;) {
    c = (ht_elem<FID>(sh, s, l) + u * c).norm().canon();
    uint32_t w[8];
    c.to_words(w);
    HtPiece a, b;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      a.w[i] = w[i];
      b.w[i] = w[4 + i];
    }
    sh.stage[s][l] = a;
    sh.stage[kHtPer + s][l] = b;
  }
  __syncthreads();
  HtPiece* dst = reinterpret_cast<HtPiece*>(out);
  const size_t base = (size_t)tile * kHtTile;
#pragma unroll
  for (uint32_t k = 0; k < 2 * kHtPer; k++) {
    const uint32_t g = l + kHtThreads * k, e = g >> 1, q = g & 1u, ll = e / kHtPer, s = e % kHtPer;
    if (base + e < n) dst[2 * base + g] = sh.stage[q * kHtPer + s][ll];
  }
}

// arena bytes of the tiled recursion for n coefficients
static inline size_t horner_tiled_need(size_t n) {
  auto pad = [](size_t b) { return (b + 255) & ~(size_t)255; };
  size_t total = 0;
  for (size_t m = n;;) {
    const size_t nt = (m + kHtTile - 1) / kHtTile;
    total += pad(nt * kHtThreads * 32) + 2 * pad(nt * 32);
    if (nt == 1) break;
    m = nt;
  }
  return total + 256;
}

}  // namespace nmx
