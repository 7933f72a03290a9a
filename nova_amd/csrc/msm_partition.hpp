// msm_partition.hpp -- hand-written bucket partition of the MSM pipeline (table mode, window width c <= 16).
//
// Replaces, on the hot path, the generic (key, value) radix sort of round 1 (rocPRIM onesweep: DigitsFn wrote
// 8 * W bytes per pair that two 8-bit sort passes re-read and re-wrote, then BoundsFn re-read the keys).  Bucket keys
// are only c - 1 <= 15 bits here, and nothing downstream needs the entries ORDERED -- the accumulate kernel needs the
// (table row | sign) words of a bucket to be contiguous, in any order.  So the partition is a two-level counting
// scatter fused with digit extraction, staged through LDS so that every global write is a contiguous run:
//
//   k_hist_hi      scalars -> signed digits (never materialised) -> LDS histogram of the HIGH key bits (<= 256 bins)
//   k_part_hi      same digits again -> LDS counting sort of the block's entries by high bits -> coalesced runs of
//                  (row | sign) words + one byte of LOW key bits into the bin's region (global cursor per bin)
//   k_hist_lo      tiles of one bin: LDS histogram of the low 7 bits -> per-bucket counts
//   k_scan_buckets exclusive scan: bucket -> [start, end)           (BoundsFn's job, without reading a sorted array)
//   k_part_lo      tiles again: LDS counting sort by low bits -> contiguous run per bucket (global cursor per bucket)
//
// Traffic at 2^20 pairs / 16 windows: 2 x 32 MB scalars + 84 MB written + 2 x 17 MB + 67 MB read + 67 MB written
// = 0.32 GB against ~0.94 GB for digits + onesweep + bounds; zero digits are dropped instead of carried to a trash
// bucket.  Order inside a bucket depends on atomics' arrival order; the bucket SUM does not (group law), and the
// affine result is canonical, so outputs stay bit-exact.
//
// The kernels use only block-level primitives (LDS atomics, __syncthreads): tests/host_emul/simt.hpp runs the very
// same bodies on the CPU, one fiber per thread, to debug them without a GPU.
#pragma once
#include "msm_kernels.hpp"

#if defined(__HIPCC__) || defined(__HIP__)
#define NMX_KERNEL __global__
#define NMX_LDS __shared__
#define NMX_SYNC() __syncthreads()
#define NMX_TID threadIdx.x
#define NMX_BDIM blockDim.x
#define NMX_BID blockIdx.x
#define NMX_GDIM gridDim.x
#define NMX_LAUNCH_BOUNDS(n) __launch_bounds__(n)
#define NMX_DEV __device__ __forceinline__
#else  // host emulation: tests/host_emul/simt.hpp must be included first
#define NMX_KERNEL
#define NMX_LDS static
#define NMX_SYNC() ::simt::syncthreads()
#define NMX_TID (::simt::tid())
#define NMX_BDIM (::simt::bdim())
#define NMX_BID (::simt::bid())
#define NMX_GDIM (::simt::gdim())
#define NMX_LAUNCH_BOUNDS(n)
#define NMX_DEV inline
#endif

namespace nmx {

// ----------------------------------------------------------------------------------------------------
// partition geometry
// ----------------------------------------------------------------------------------------------------
struct PartShape {
  uint32_t LB, HB;    // low / high key bits: key = bucket index in [0, 2^(c-1)), LB = min(c - 1, 7), HB <= 8
  uint32_t nlo, nhi;  // 2^LB, 2^HB
  uint32_t bs1;       // threads (= scalars) per block iteration of the first level: bs1 * W <= kStageCap
  uint32_t grid1;     // blocks of the first level (grid-stride over chunks of bs1 scalars)
  uint32_t tiles_cap; // upper bound on second-level tiles
};
static constexpr uint32_t kStageCap = 16384;  // entries staged in LDS per first-level chunk (64 KiB + 16 KiB)
static constexpr uint32_t kTile = 8192;       // entries per second-level tile (32 KiB of LDS)
static constexpr uint32_t kTileThreads = 1024;
static constexpr uint32_t kTilePer = kTile / kTileThreads;

inline bool partition_supported(const MsmShape& sh, bool table_mode) {
  return table_mode && sh.WB == 1 && sh.c >= 2 && sh.c <= 16 && sh.W <= 64;
}
inline PartShape make_part_shape(const MsmShape& sh) {
  PartShape p;
  const uint32_t kb = sh.c - 1;
  p.LB = kb < 7 ? kb : 7;
  p.HB = kb - p.LB;
  p.nlo = 1u << p.LB;
  p.nhi = 1u << p.HB;
  uint32_t bs = (kStageCap / sh.W) & ~63u;
  if (bs > 1024) bs = 1024;
  if (bs < 64) bs = 64;
  p.bs1 = bs;
  const uint32_t chunks = (sh.n + bs - 1) / bs;
  p.grid1 = chunks < 512 ? chunks : 512;
  p.tiles_cap = (uint32_t)(((uint64_t)sh.n * sh.W) / kTile) + p.nhi + 1;
  return p;
}

// exclusive prefix sums of a[0..n) into out[0..n] (out[n] = total), n <= 1024, by the first n threads of the block;
// every thread of the block must call it (barriers inside).  A plain loop per thread: n is at most a few hundred and
// the reads are LDS broadcasts.
NMX_DEV void block_excl_scan(const uint32_t* a, uint32_t* out, uint32_t n) {
  for (uint32_t i = NMX_TID; i <= n; i += NMX_BDIM) {
    uint32_t acc = 0;
    for (uint32_t j = 0; j < i; j++) acc += a[j];
    out[i] = acc;
  }
  NMX_SYNC();
}
// largest b in [0, n) with base[b] <= s, for base[] nondecreasing, base[0] = 0 <= s < base[n]
NMX_DEV uint32_t find_bin(const uint32_t* base, uint32_t n, uint32_t s) {
  uint32_t lo = 0, hi = n;  // invariant: base[lo] <= s < base[hi]
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (base[mid] <= s) lo = mid;
    else hi = mid;
  }
  return lo;
}

// ----------------------------------------------------------------------------------------------------
// level 1
// ----------------------------------------------------------------------------------------------------
template <int SFID> struct PartArgs {
  DigitSrc<SFID> src;
  PartShape ps;
  uint32_t* hist_hi;    // [256]  entries per high bin                    (zeroed)
  uint32_t* cur_hi;     // [256]  scatter cursors per high bin            (zeroed)
  uint32_t* ent_val;    // [n * W] first-level output: (table row | sign << 31), grouped by high bin
  uint8_t* ent_lo;      // [n * W] low key bits of the same entries
  uint32_t* bucket_cnt; // [nbuckets + 1] entries per bucket               (zeroed)
  uint32_t* bucket_cur; // [nbuckets]     scatter cursors per bucket       (zeroed)
  uint32_t* start;      // [nbuckets + 1]
  uint32_t* end;        // [nbuckets + 1]
  uint32_t* vals;       // [n * W] final: words of bucket k at [start[k], end[k])
  uint32_t* total_out;  // number of non-zero digits of the whole MSM
};

template <int SFID> NMX_KERNEL void NMX_LAUNCH_BOUNDS(1024) k_hist_hi(PartArgs<SFID> a) {
  NMX_LDS uint32_t cnt[256];
  const uint32_t t = NMX_TID, bs = NMX_BDIM;
  for (uint32_t j = t; j < 256; j += bs) cnt[j] = 0;
  NMX_SYNC();
  const MsmShape& sh = a.src.sh;
  for (uint32_t base = NMX_BID * bs; base < sh.n; base += NMX_GDIM * bs) {
    const uint32_t i = base + t;
    if (i < sh.n) {
      uint32_t s[9], bi;
      if (a.src.load(i, s, bi, true)) {
        uint32_t carry = 0;
        for (uint32_t w = 0; w < sh.W; w++) {
          uint32_t d, neg;
          a.src.digit(s, w, carry, d, neg);
          if (d) nmx_atomic_add(&cnt[(d - 1) >> a.ps.LB], 1u);
        }
      }
    }
  }
  NMX_SYNC();
  for (uint32_t j = t; j < a.ps.nhi; j += bs)
    if (cnt[j]) nmx_atomic_add(&a.hist_hi[j], cnt[j]);
}

template <int SFID> NMX_KERNEL void NMX_LAUNCH_BOUNDS(1024) k_part_hi(PartArgs<SFID> a) {
  NMX_LDS uint32_t stage_val[kStageCap];
  NMX_LDS uint8_t stage_lo[kStageCap];
  NMX_LDS uint32_t binstart[257], cnt[256], lbase[257], gbase[256], cur[256];
  const uint32_t t = NMX_TID, bs = NMX_BDIM;
  const MsmShape& sh = a.src.sh;
  const uint32_t nhi = a.ps.nhi, LB = a.ps.LB, lomask = a.ps.nlo - 1u;
  for (uint32_t j = t; j < 256; j += bs) cnt[j] = j < nhi ? a.hist_hi[j] : 0;
  NMX_SYNC();
  block_excl_scan(cnt, binstart, nhi);  // where each high bin's region starts in ent_val / ent_lo
  for (uint32_t base = NMX_BID * bs; base < sh.n; base += NMX_GDIM * bs) {
    for (uint32_t j = t; j < 256; j += bs) cnt[j] = 0, cur[j] = 0;
    NMX_SYNC();
    const uint32_t i = base + t;
    uint32_t s[9], bi = 0;
    const bool live = i < sh.n && a.src.load(i, s, bi, false);
    if (live) {  // phase A: count this chunk's entries per bin
      uint32_t carry = 0;
      for (uint32_t w = 0; w < sh.W; w++) {
        uint32_t d, neg;
        a.src.digit(s, w, carry, d, neg);
        if (d) nmx_atomic_add(&cnt[(d - 1) >> LB], 1u);
      }
    }
    NMX_SYNC();
    block_excl_scan(cnt, lbase, nhi);
    for (uint32_t j = t; j < nhi; j += bs)
      if (cnt[j]) gbase[j] = binstart[j] + nmx_atomic_add(&a.cur_hi[j], cnt[j]);  // reserve the bin's run
    NMX_SYNC();
    if (live) {  // phase B: the same digits again, now placed: LDS slot = bin's local base + arrival rank
      uint32_t carry = 0;
      for (uint32_t w = 0; w < sh.W; w++) {
        uint32_t d, neg;
        a.src.digit(s, w, carry, d, neg);
        if (d) {
          const uint32_t key = d - 1, bin = key >> LB;
          const uint32_t slot = lbase[bin] + nmx_atomic_add(&cur[bin], 1u);
          stage_val[slot] = (w * a.src.pre_stride + bi) | (neg << 31);
          stage_lo[slot] = (uint8_t)(key & lomask);
        }
      }
    }
    NMX_SYNC();
    const uint32_t tot = lbase[nhi];
    for (uint32_t sl = t; sl < tot; sl += bs) {  // consecutive threads -> consecutive addresses inside a bin's run
      const uint32_t bin = find_bin(lbase, nhi, sl);
      const uint32_t g = gbase[bin] + (sl - lbase[bin]);
      a.ent_val[g] = stage_val[sl];
      a.ent_lo[g] = stage_lo[sl];
    }
    NMX_SYNC();
  }
}

// ----------------------------------------------------------------------------------------------------
// level 2: tiles of <= kTile entries, each inside one high bin
// ----------------------------------------------------------------------------------------------------
// tile -> (bin, first entry, length); false past the last tile.  Every thread of the block calls it.
template <int SFID> NMX_DEV bool tile_of_block(const PartArgs<SFID>& a, uint32_t* binstart, uint32_t* tilestart,
                                               uint32_t* scratch, uint32_t& bin, uint32_t& first, uint32_t& len) {
  const uint32_t t = NMX_TID, nhi = a.ps.nhi;
  if (t < 256) scratch[t] = t < nhi ? a.hist_hi[t] : 0;
  NMX_SYNC();
  block_excl_scan(scratch, binstart, nhi);
  if (t < 256) scratch[t] = t < nhi ? (scratch[t] + kTile - 1) / kTile : 0;
  NMX_SYNC();
  block_excl_scan(scratch, tilestart, nhi);
  const uint32_t tile = NMX_BID;
  if (tile >= tilestart[nhi]) return false;  // block-uniform
  bin = find_bin(tilestart, nhi, tile);
  const uint32_t j = tile - tilestart[bin], size = binstart[bin + 1] - binstart[bin];
  first = binstart[bin] + j * kTile;
  len = size - j * kTile < kTile ? size - j * kTile : kTile;
  return true;
}

template <int SFID> NMX_KERNEL void NMX_LAUNCH_BOUNDS(1024) k_hist_lo(PartArgs<SFID> a) {
  NMX_LDS uint32_t binstart[257], tilestart[257], scratch[256], cnt[128];
  uint32_t bin, first, len;
  if (!tile_of_block(a, binstart, tilestart, scratch, bin, first, len)) return;
  const uint32_t t = NMX_TID;
  if (t < 128) cnt[t] = 0;
  NMX_SYNC();
  for (uint32_t j = t; j < len; j += NMX_BDIM) nmx_atomic_add(&cnt[a.ent_lo[first + j]], 1u);
  NMX_SYNC();
  if (t < a.ps.nlo && cnt[t]) nmx_atomic_add(&a.bucket_cnt[(bin << a.ps.LB) + t], cnt[t]);
}

// bucket -> [start, end): exclusive scan of the bucket counts (one block; nbuckets <= 2^15)
template <int SFID> NMX_KERNEL void NMX_LAUNCH_BOUNDS(1024) k_scan_buckets(PartArgs<SFID> a) {
  NMX_LDS uint32_t part[1024], pre[1025];
  const uint32_t t = NMX_TID, bs = NMX_BDIM, nb = a.src.sh.nbuckets;
  const uint32_t per = (nb + bs - 1) / bs, lo = t * per, hi = lo + per < nb ? lo + per : nb;
  uint32_t sum = 0;
  for (uint32_t k = lo; k < hi; k++) sum += a.bucket_cnt[k];
  part[t] = sum;
  NMX_SYNC();
  block_excl_scan(part, pre, bs);
  uint32_t run = pre[t];
  for (uint32_t k = lo; k < hi; k++) {
    const uint32_t c = a.bucket_cnt[k];
    a.start[k] = run;
    run += c;
    a.end[k] = run;
  }
  if (t == 0) {
    a.start[nb] = pre[bs];  // the (empty) trash slot of the rocPRIM layout: keeps start[nbuckets] / end[nbuckets] defined
    a.end[nb] = pre[bs];
    *a.total_out = pre[bs];
  }
}

template <int SFID> NMX_KERNEL void NMX_LAUNCH_BOUNDS(1024) k_part_lo(PartArgs<SFID> a) {
  NMX_LDS uint32_t stage[kTile];
  NMX_LDS uint32_t binstart[257], tilestart[257], scratch[256], cnt[128], lbase[129], gbase[128];
  uint32_t bin, first, len;
  if (!tile_of_block(a, binstart, tilestart, scratch, bin, first, len)) return;
  const uint32_t t = NMX_TID, bs = NMX_BDIM, nlo = a.ps.nlo;
  if (t < 128) cnt[t] = 0;
  NMX_SYNC();
  uint32_t v[kTilePer], lo[kTilePer], pos[kTilePer];
#pragma unroll
  for (uint32_t j = 0; j < kTilePer; j++) {
    const uint32_t e = j * bs + t;
    if (e < len) {
      v[j] = a.ent_val[first + e];
      lo[j] = a.ent_lo[first + e];
      pos[j] = nmx_atomic_add(&cnt[lo[j]], 1u);  // arrival rank inside the tile's sub-bin
    }
  }
  NMX_SYNC();
  block_excl_scan(cnt, lbase, nlo);
  if (t < nlo && cnt[t]) {
    const uint32_t k = (bin << a.ps.LB) + t;
    gbase[t] = a.start[k] + nmx_atomic_add(&a.bucket_cur[k], cnt[t]);  // this tile's run inside bucket k
  }
#pragma unroll
  for (uint32_t j = 0; j < kTilePer; j++)
    if (j * bs + t < len) stage[lbase[lo[j]] + pos[j]] = v[j];
  NMX_SYNC();
  for (uint32_t sl = t; sl < len; sl += bs) {
    const uint32_t b = find_bin(lbase, nlo, sl);
    a.vals[gbase[b] + (sl - lbase[b])] = stage[sl];
  }
}

}  // namespace nmx
