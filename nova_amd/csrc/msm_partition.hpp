// msm_partition.hpp -- hand-written bucket partition of the MSM pipeline (keys with window tables, and since round 4 plain keys too).
//
// Replaces, on the hot path, the generic (key, value) radix sort of round 1 (rocPRIM onesweep: DigitsFn wrote
// 8 * W bytes per pair that two 8-bit sort passes re-read and re-wrote, then BoundsFn re-read the keys).  Bucket keys
// are only c - 1 <= 15 bits here, and nothing downstream needs the entries ORDERED -- the accumulate kernel needs the
// (table row | sign) words of a bucket to be contiguous, in any order.  So the partition is a two-level counting
// scatter fused with digit extraction, staged through LDS so that every global write is a contiguous run:
//
//   k_hist_hi   scalars -> signed digits (never materialised) -> LDS histogram of the HIGH key bits (<= 256 bins)
//   k_tiles     one block: bin regions (16-entry aligned) of the intermediate arrays, bin regions of the final array,
//               tile table of level 2, start / end of the buckets of empty bins
//   k_part_hi   the same digits (kept in registers between the counting and the placing phase) -> LDS counting sort of
//               the block's entries by high bits -> coalesced runs of (row | sign) words + one byte of LOW key bits
//               into the bin's region (one global cursor per bin)
//   k_hist_lo   tiles of one bin: LDS histogram of the low 7 bits -> per-bucket counts
//   k_part_lo   tiles again: bucket offsets from a bin-local scan of those counts (no global scan kernel: a bin's buckets
//               occupy the bin's region of the final array), LDS counting sort by low bits -> contiguous run per
//               bucket (one global cursor per bucket); the first tile of a bin publishes its buckets' [start, end)
//
// Traffic at 2^20 pairs / 16 windows (c = 16; the c = 17 tables of 2^20-point keys have 15): 2 x 32 MB scalars + 84 MB written + 17 MB + 84 MB read + 67 MB written
// = 0.32 GB against ~0.94 GB for digits + onesweep + bounds; zero digits are dropped instead of carried to a trash
// bucket.  Order inside a bucket depends on atomics' arrival order; the bucket SUM does not (group law), and the
// affine result is canonical, so outputs stay bit-exact.
//
// The window width is a template parameter of the level-1 kernels (8 / 15 / 16: the widths the tables are built
// with; 0 = any width at run time): with constant bit positions the scalar words stay in registers -- indexed
// dynamically they are promoted to LDS (36 KB per block) -- and the digit loop unrolls to ~10 instructions a window.
//
// Two geometries (template parameter BIG): keys of up to 16 bits (c <= 17: every key below 2^22 points) use 256 x 128 bins
// (256 x 256 for 16 bits), 12288-entry chunks, 8192-entry tiles, one byte of low bits per entry and two first-level blocks
// per CU; keys of up to 20 bits (the c = 20 tables of keys >= 2^22 points, the bucket sets of fused batches) use 1024 x 512
// bins (1024 x 1024 for 20 bits), 16384-entry chunks and tiles, two bytes of low bits, one block per CU -- shorter runs per bin (64 B), still one
// pass per level.
//
// Apart from the wave scan (device only; a plain loop elsewhere) the kernels use block-level primitives only (LDS
// atomics, __syncthreads): tests/host_emul/simt.hpp runs the same bodies on the CPU, one fiber per thread.
#pragma once
#include <string.h>

#include <type_traits>

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
// geometry
// ----------------------------------------------------------------------------------------------------
struct PartShape {
  uint32_t LB, HB;     // low / high key bits: key = bucket index in [0, 2^(c-1))
  uint32_t nlo, nhi;   // 2^LB, 2^HB
  uint32_t big;        // 1: the wide-key geometry (PartCfg<true>)
  uint32_t bs1;        // threads (= scalars) per block iteration of the first level: bs1 * W <= stage capacity
  uint32_t grid1;      // blocks of the first level (grid-stride over chunks of bs1 scalars)
  uint32_t tiles_cap;  // upper bound on second-level tiles
  uint32_t ent_cap;    // entries of the intermediate arrays (bin regions are 16-entry aligned, one tile of slack)
};
template <bool BIG> struct PartCfg {
  static constexpr uint32_t kMaxHi = BIG ? 1024 : 256;    // first-level bins
  static constexpr uint32_t kMaxLo = BIG ? 1024 : 256;    // buckets per bin (narrow: 128 up to 15 key bits, 256 for 16; wide: 512, 1024 for 20)
  static constexpr uint32_t kStage = BIG ? 16384 : 12288; // entries staged in LDS per first-level chunk
  static constexpr uint32_t kTile = BIG ? 16384 : 8192;   // entries per second-level tile
  static constexpr uint32_t kTilePer = kTile / 1024;      // consecutive entries per thread of a tile
  using KeyT = typename std::conditional<BIG, uint32_t, uint16_t>::type;  // staged key
  using LoT = typename std::conditional<BIG, uint16_t, uint8_t>::type;    // low key bits between the levels
};
static constexpr uint32_t kTileThreads = 1024;
static constexpr uint32_t kBinAlign = 16;
static constexpr uint32_t kTabStride = 1025;  // tab = 3 arrays of nhi + 1 <= 1025 words

// Geometry for a shape; false when the hand-written partition does not cover it (the generic sort path runs).
// Keys are bucket indices in [0, WB * M).  Table mode: c - 1 bits for one vector, up to 20 with the bucket sets of a fused
// batch (WB a power of two).  Plain keys (round 4: first / second sight of a cached array, IPA's per-round keys, one-shot
// uploads, keys whose tables did not fit -- /root/reference/src/provider/msm.rs:577-661 is the per-window bucket method they
// replace): WB = W bucket sets, key = w * M + |d| - 1, the same two-level counting scatter over ceil(log2(W * M)) key bits; the
// bins past the last bucket stay empty.
inline bool make_part_shape(const MsmShape& sh, bool table_mode, PartShape* out) {
  if (!(sh.c >= 2 && sh.c <= 20 && sh.WB >= 1)) return false;
  if (table_mode && (sh.WB & (sh.WB - 1)) != 0) return false;
  uint32_t kb = 0;
  while (((uint64_t)1 << kb) < (uint64_t)sh.WB * sh.M) kb++;
  if (kb > 20) return false;
  PartShape p;
  p.big = kb > 16 ? 1u : 0u;
  p.LB = p.big ? (kb == 20 ? 10u : 9u) : (kb < 7 ? kb : (kb == 16 ? 8u : 7u));
  p.HB = kb - p.LB;
  p.nlo = 1u << p.LB;
  p.nhi = 1u << p.HB;
  const uint32_t stage = p.big ? PartCfg<true>::kStage : PartCfg<false>::kStage;
  const uint32_t tile = p.big ? PartCfg<true>::kTile : PartCfg<false>::kTile;
  // one scalar per thread and chunk: its W digits must fit the LDS stage; 256 threads is the smallest first-level block,
  // and the block also scans the nhi bins (block_excl_scan needs a thread per bin)
  uint32_t bs = (stage / sh.W) & ~63u;
  if (bs > 1024) bs = 1024;
  if (bs < (table_mode ? 256u : 64u) || bs < p.nhi) return false;  // (tiny plain MSMs, c = 3: 85 windows -> 128-thread chunks)
  p.bs1 = bs;
  const uint32_t chunks = (sh.n + bs - 1) / bs;
  p.grid1 = chunks < 1024 ? chunks : 1024;
  p.ent_cap = (uint32_t)((uint64_t)sh.n * sh.W) + kBinAlign * p.nhi + tile;
  p.tiles_cap = p.ent_cap / tile + p.nhi + 1;
  *out = p;
  return true;
}
inline bool partition_supported(const MsmShape& sh, bool table_mode) {
  PartShape p;
  return make_part_shape(sh, table_mode, &p);
}
inline PartShape make_part_shape(const MsmShape& sh, bool table_mode) {
  PartShape p{};
  (void)make_part_shape(sh, table_mode, &p);
  return p;
}

// Exclusive prefix sums of a[0..n) into out[0..n] (out[n] = total), n <= 1024; every thread of the block must call it
// (barriers inside).  Device: wave scan + one LDS hop (needs blockDim >= n); elsewhere: a loop per element.
NMX_DEV void block_excl_scan(const uint32_t* a, uint32_t* out, uint32_t n, uint32_t* wtot /* LDS, 16 words */) {
#if defined(__HIP_DEVICE_COMPILE__)
  const uint32_t t = NMX_TID, lane = t & 63u, wv = t >> 6;
  const uint32_t v = t < n ? a[t] : 0u;
  uint32_t inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t u = __shfl_up(inc, d);
    if (lane >= (uint32_t)d) inc += u;
  }
  if (lane == 63) wtot[wv] = inc;
  NMX_SYNC();
  uint32_t pre = 0;
  for (uint32_t w = 0; w < wv; w++) pre += wtot[w];
  if (t < n) out[t] = pre + inc - v;
  if (t + 1 == n) out[n] = pre + inc;
  if (n == 0 && t == 0) out[0] = 0;
  NMX_SYNC();
#else
  (void)wtot;
  for (uint32_t i = NMX_TID; i <= n; i += NMX_BDIM) {
    uint32_t acc = 0;
    for (uint32_t j = 0; j < i; j++) acc += a[j];
    out[i] = acc;
  }
  NMX_SYNC();
#endif
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

// LDS counter updates by all active lanes of a wave.  Skewed scalars (all equal, 0 / r-1, 0 / 1 ...) send every lane
// of a wave to the SAME counter, and same-address LDS atomics serialise lane by lane; when the active lanes agree on
// the counter one of them adds the population count instead (measured on all-equal scalars at 2^20: first level
// 0.14 + 0.33 ms without this).  Mixed targets -- the random case -- take the plain atomic.
NMX_DEV void lds_count(uint32_t* cnt, uint32_t bin) {
#if defined(__HIP_DEVICE_COMPILE__)
  const uint32_t first = (uint32_t)__builtin_amdgcn_readfirstlane((int)bin);
  if (__all(bin == first)) {
    const unsigned long long m = __ballot(1);
    if (__lane_id() == (uint32_t)(__ffsll((long long)m) - 1)) atomicAdd(&cnt[first], (uint32_t)__popcll(m));
    return;
  }
#endif
  nmx_atomic_add(&cnt[bin], 1u);
}
// same, returning this lane's arrival rank
NMX_DEV uint32_t lds_rank(uint32_t* cur, uint32_t bin) {
#if defined(__HIP_DEVICE_COMPILE__)
  const uint32_t first = (uint32_t)__builtin_amdgcn_readfirstlane((int)bin);
  if (__all(bin == first)) {
    const unsigned long long m = __ballot(1);
    const uint32_t lane = __lane_id(), leader = (uint32_t)(__ffsll((long long)m) - 1);
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(&cur[first], (uint32_t)__popcll(m));
    base = (uint32_t)__shfl((int)base, (int)leader);
    return base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
  }
#endif
  return nmx_atomic_add(&cur[bin], 1u);
}

// buffers shared by all partition kernels
struct PartBufs {
  PartShape ps;
  uint32_t nbuckets;
  uint32_t* hist_hi;     // [nhi]  entries per high bin                                  (zeroed)
  uint32_t* cur_hi;      // [nhi]  scatter cursors per high bin                          (zeroed)
  uint32_t* tab;         // [3 x kTabStride] ent_base | binstart | tilestart  (k_tiles)
  uint32_t* ent_val;     // [ent_cap] level-1 output: (table row | sign << 31), grouped by high bin
  void* ent_lo;          // [ent_cap] low key bits of the same entries (PartCfg::LoT: one or two bytes)
  uint32_t* bucket_cnt;  // [nbuckets] entries per bucket                                (zeroed)
  uint32_t* bucket_cur;  // [nbuckets] scatter cursors per bucket                        (zeroed)
  uint32_t* start;       // [nbuckets + 1]
  uint32_t* end;         // [nbuckets + 1]
  uint32_t* vals;        // [n * W] final: words of bucket k at [start[k], end[k])
  uint32_t* total_out;   // number of non-zero digits of the whole MSM
  // ONE high bin (keys of up to 7 bits: a single vector over c = 8 tables -- every key below 2^14 points, the secondary circuit's
  // commitments, the narrow prefix tables): the counting pass has nothing to decide -- the bin's region starts at 0 and its size is
  // what the placing pass's cursor ends at.  The pipeline then runs k_part_hi FIRST (reporting range errors itself), k_tiles reads
  // the count from cur_hi, and k_hist_hi is not launched at all: one dependent launch less on MSMs that are pure latency.
  uint32_t single_bin = 0;
};
template <int SFID> struct PartArgs {
  DigitSrc<SFID> src;
  PartBufs b;
};

// ----------------------------------------------------------------------------------------------------
// digits of one scalar, window width C at compile time (C = 0: run-time width, src.sh.c)
// ----------------------------------------------------------------------------------------------------
template <int C> struct WinMax {
  static constexpr uint32_t value = C ? (256 + C - 1) / C : 64;
};
// (hi:lo >> off)[31:0], off < 32
NMX_HD uint32_t funnel_shr(uint32_t hi, uint32_t lo, uint32_t off) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_alignbit(hi, lo, off);
#else
  return off ? (lo >> off) | (hi << (32 - off)) : lo;
#endif
}
// calls f(w, |d|, neg) for every window with a non-zero digit
template <int SFID, int C, class Fn> NMX_HD void for_each_digit(const DigitSrc<SFID>& src, const uint32_t (&s)[9], Fn&& f) {
  const MsmShape& sh = src.sh;
  uint32_t carry = 0;
  if constexpr (C == 0) {
    // run-time width: walk the WORDS (constant indices: the scalar stays in registers) and peel windows off a 64-bit bit
    // buffer.  Indexing s[] by a run-time word number -- src.digit -- kept the array in memory: 36 KB of LDS per block in
    // k_hist_hi, scratch in k_part_hi, two dependent reads per digit.  c <= 20, so the buffer never holds more than 51 bits.
    const uint32_t c = sh.c, mask = (1u << c) - 1u, half = sh.M;
    uint64_t acc = 0;
    uint32_t have = 0, w = 0;
#pragma unroll
    for (int j = 0; j < 9; j++) {  // s[8] = 0: the padding the top window reads
      acc |= (uint64_t)s[j] << have;
      have += 32;
      while (have >= c && w < sh.W) {
        uint32_t d = ((uint32_t)acc & mask) + carry, neg = 0;
        acc >>= c;
        have -= c;
        if (d > half) {
          d = (1u << c) - d;
          neg = 1;
          carry = 1;
        } else {
          carry = 0;
        }
        if (d) f(w, d, neg);
        w++;
      }
    }
  } else {
    constexpr uint32_t mask = (1u << C) - 1u, half = 1u << (C - 1);
#pragma unroll
    for (uint32_t w = 0; w < WinMax<C>::value; w++) {
      if (w < sh.W) {
        const uint32_t bit = w * C, word = bit >> 5, off = bit & 31;  // constants after unrolling
        // a 32-bit funnel shift (v_alignbit_b32), NOT a 64-bit value built from two words: the optimiser turns that into one
        // 64-bit load of the ARRAY, and an array read that way is never scalarised -- it stayed in memory, promoted to LDS
        // (36 KB per block in k_hist_hi) or spilled to scratch (k_part_hi), two dependent memory reads per digit (round 4,
        // found in the kernels' ISA; same-box A/B at 2^20: k_hist_hi + k_tiles 53 -> 43 us, the scatter passes 138 -> 123 us,
        // profiles/r04_msm_2p20/digit_ab.txt)
        uint32_t d = (funnel_shr(s[word < 8 ? word + 1 : 8], s[word < 8 ? word : 8], off) & mask) + carry;  // s[8] = 0
        uint32_t neg = 0;
        if (d > half) {
          d = (1u << C) - d;
          neg = 1;
          carry = 1;
        } else {
          carry = 0;
        }
        if (d) f(w, d, neg);
      }
    }
  }
}

// ----------------------------------------------------------------------------------------------------
// level 1
// ----------------------------------------------------------------------------------------------------
template <int SFID, int C, bool BIG> NMX_KERNEL void NMX_LAUNCH_BOUNDS(1024) k_hist_hi(PartArgs<SFID> a) {
  using Cfg = PartCfg<BIG>;
  NMX_LDS uint32_t cnt[Cfg::kMaxHi];
  const uint32_t t = NMX_TID, bs = NMX_BDIM, LB = a.b.ps.LB;
  for (uint32_t j = t; j < Cfg::kMaxHi; j += bs) cnt[j] = 0;
  NMX_SYNC();
  const uint32_t n = a.src.sh.n;
  for (uint32_t base = NMX_BID * bs; base < n; base += NMX_GDIM * bs) {
    const uint32_t i = base + t;
    if (i < n) {
      uint32_t s[9], bi, kbase;
      if (a.src.load(i, s, bi, kbase, true))
        for_each_digit<SFID, C>(a.src, s, [&](uint32_t w, uint32_t d, uint32_t) { lds_count(cnt, (a.src.key_base(w, kbase) + d - 1) >> LB); });
    }
  }
  NMX_SYNC();
  for (uint32_t j = t; j < a.b.ps.nhi; j += bs)
    if (cnt[j]) nmx_atomic_add(&a.b.hist_hi[j], cnt[j]);
}

// One block of 1024 threads.  tab[0 .. nhi] = ent_base (bin regions of ent_val / ent_lo, 16-entry aligned),
// tab[kTabStride ..] = binstart (bin regions of the final array = exclusive scan of the bin sizes),
// tab[2 * kTabStride ..] = tilestart.  (The SFID-independent kernels are templates only to get inline linkage
// across the curve TUs.)
template <bool BIG> NMX_KERNEL void NMX_LAUNCH_BOUNDS(1024) k_tiles(PartBufs b) {
  using Cfg = PartCfg<BIG>;
  NMX_LDS uint32_t h[1024], al[1024], tl[1024], o1[1025], o2[1025], o3[1025], wtot[16], any_empty;
  const uint32_t t = NMX_TID, nhi = b.ps.nhi;
  const uint32_t c = t < nhi ? b.hist_hi[t] : 0;
  if (t == 0) any_empty = 0;
  NMX_SYNC();
  if (t < nhi && c == 0) any_empty = 1;  // (same value from every writer)
  h[t] = c;
  al[t] = (c + kBinAlign - 1) & ~(kBinAlign - 1);
  tl[t] = (c + Cfg::kTile - 1) / Cfg::kTile;
  NMX_SYNC();
  block_excl_scan(al, o1, nhi, wtot);
  block_excl_scan(h, o2, nhi, wtot);
  block_excl_scan(tl, o3, nhi, wtot);
  for (uint32_t j = t; j <= nhi; j += NMX_BDIM) {
    b.tab[j] = o1[j];
    b.tab[kTabStride + j] = o2[j];
    b.tab[2 * kTabStride + j] = o3[j];
  }
  // No tile will ever visit an empty bin: its buckets are empty, placed at the bin's offset.  The empty bins are listed (a
  // scan of their flags) and the block walks THEIR buckets only, consecutive threads on consecutive buckets (a thread per bin
  // writing its nlo buckets one by one was 0.15 ms of uncoalesced stores for the sparse bucket sets of a fused batch of short
  // vectors; a serial walk over the bins cost 18 us of dependent LDS reads on every call; a walk over ALL buckets whenever one
  // bin is empty would be paid by every plain-key MSM, whose top window never fills its upper bins).
  if (any_empty) {  // block-uniform; uniform random scalars over window tables: no empty bin, nothing to do
    al[t] = (t < nhi && c == 0) ? 1u : 0u;
    NMX_SYNC();
    block_excl_scan(al, o1, nhi, wtot);
    if (t < nhi && c == 0) tl[o1[t]] = t;
    NMX_SYNC();
    const uint32_t ne = o1[nhi], LB = b.ps.LB, nlo = b.ps.nlo;
    for (uint32_t idx = t; idx < (ne << LB); idx += NMX_BDIM) {
      const uint32_t bin = tl[idx >> LB], k = (bin << LB) + (idx & (nlo - 1u));
      if (k < b.nbuckets) {  // plain keys: W * M need not fill the last bins
        b.start[k] = o2[bin];
        b.end[k] = o2[bin];
      }
    }
  }
  if (t == 0) {
    b.start[b.nbuckets] = o2[nhi];  // the (empty) trash slot of the generic layout stays defined
    b.end[b.nbuckets] = o2[nhi];
    *b.total_out = o2[nhi];
  }
}

template <int SFID, int C, bool BIG> NMX_KERNEL void NMX_LAUNCH_BOUNDS(1024) k_part_hi(PartArgs<SFID> a) {
  using Cfg = PartCfg<BIG>;
  using KeyT = typename Cfg::KeyT;
  using LoT = typename Cfg::LoT;
  NMX_LDS uint32_t stage_val[Cfg::kStage];
  NMX_LDS KeyT stage_key[Cfg::kStage];
  NMX_LDS uint32_t ent_base[Cfg::kMaxHi], cnt[Cfg::kMaxHi], lbase[Cfg::kMaxHi + 1], gbase[Cfg::kMaxHi], cur[Cfg::kMaxHi], wtot[16];
  const uint32_t t = NMX_TID, bs = NMX_BDIM;
  const uint32_t n = a.src.sh.n, nhi = a.b.ps.nhi, LB = a.b.ps.LB, lomask = a.b.ps.nlo - 1u;
  LoT* ent_lo = static_cast<LoT*>(a.b.ent_lo);
  for (uint32_t j = t; j < nhi; j += bs) ent_base[j] = a.b.tab[j];
  for (uint32_t base = NMX_BID * bs; base < n; base += NMX_GDIM * bs) {
    for (uint32_t j = t; j < Cfg::kMaxHi; j += bs) cnt[j] = 0, cur[j] = 0;
    NMX_SYNC();
    const uint32_t i = base + t;
    uint32_t s[9], bi = 0, kbase = 0;
    const bool live = i < n && a.src.load(i, s, bi, kbase, a.b.single_bin != 0);
    // phase A: this chunk's entries per bin.  With a compile-time width the digits stay in registers for phase B
    // (key | neg << 31, all ones = none); at run-time width they are extracted again.
    uint32_t dig[C ? WinMax<C>::value : 1];
    if constexpr (C != 0) {
#pragma unroll
      for (uint32_t w = 0; w < WinMax<C>::value; w++) dig[w] = 0xffffffffu;
    }
    if (live)
      for_each_digit<SFID, C>(a.src, s, [&](uint32_t w, uint32_t d, uint32_t neg) {
        const uint32_t key = a.src.key_base(w, kbase) + d - 1;
        lds_count(cnt, key >> LB);
        if constexpr (C != 0) dig[w] = key | (neg << 31);
      });
    NMX_SYNC();
    block_excl_scan(cnt, lbase, nhi, wtot);
    for (uint32_t j = t; j < nhi; j += bs)
      if (cnt[j]) gbase[j] = ent_base[j] + nmx_atomic_add(&a.b.cur_hi[j], cnt[j]);  // reserve the bin's run
    // phase B: placed -- LDS slot = bin's local base + arrival rank
    if (live) {
      auto place = [&](uint32_t w, uint32_t key, uint32_t neg) {
        const uint32_t bin = key >> LB;
        const uint32_t slot = lbase[bin] + lds_rank(cur, bin);
        stage_val[slot] = (w * a.src.pre_stride + bi) | (neg << 31);
        stage_key[slot] = (KeyT)key;
      };
      if constexpr (C != 0) {
#pragma unroll
        for (uint32_t w = 0; w < WinMax<C>::value; w++)
          if (dig[w] != 0xffffffffu) place(w, dig[w] & 0x7fffffffu, dig[w] >> 31);
      } else {
        for_each_digit<SFID, C>(a.src, s, [&](uint32_t w, uint32_t d, uint32_t neg) { place(w, a.src.key_base(w, kbase) + d - 1, neg); });
      }
    }
    NMX_SYNC();
    const uint32_t tot = lbase[nhi];
    for (uint32_t sl = t; sl < tot; sl += bs) {  // consecutive threads -> consecutive addresses inside a bin's run
      const uint32_t key = stage_key[sl], bin = key >> LB;
      const uint32_t g = gbase[bin] + (sl - lbase[bin]);
      a.b.ent_val[g] = stage_val[sl];
      ent_lo[g] = (LoT)(key & lomask);
    }
    NMX_SYNC();
  }
}

// ----------------------------------------------------------------------------------------------------
// level 2: tiles of <= kTile entries, each inside one high bin (tile starts are 16-entry aligned)
// ----------------------------------------------------------------------------------------------------
// tile -> (bin, tile index inside the bin, first entry in ent_*, length); false past the last tile (block-uniform).
template <bool BIG>
NMX_DEV bool tile_of_block(const PartBufs& b, uint32_t* tilestart /* LDS kMaxHi + 1 */, uint32_t& bin, uint32_t& j,
                           uint32_t& first, uint32_t& len) {
  constexpr uint32_t T = PartCfg<BIG>::kTile;
  const uint32_t nhi = b.ps.nhi;
  for (uint32_t k = NMX_TID; k <= nhi; k += NMX_BDIM) tilestart[k] = b.tab[2 * kTabStride + k];
  NMX_SYNC();
  const uint32_t tile = NMX_BID;
  if (tile >= tilestart[nhi]) return false;
  bin = find_bin(tilestart, nhi, tile);
  j = tile - tilestart[bin];
  const uint32_t size = b.tab[kTabStride + bin + 1] - b.tab[kTabStride + bin];
  first = b.tab[bin] + j * T;
  len = size - j * T < T ? size - j * T : T;
  return true;
}
// this thread's kTilePer consecutive low-bit values: aligned 8- / 32-byte loads (values past the tile's length may be read
// -- the arrays carry a tile of slack -- and are masked by the callers)
template <bool BIG> NMX_DEV void load_lo(const void* base, size_t e, uint32_t (&lo)[PartCfg<BIG>::kTilePer]) {
  if constexpr (!BIG) {
    uint64_t w;
    memcpy(&w, __builtin_assume_aligned(static_cast<const uint8_t*>(base) + e, 8), 8);
#pragma unroll
    for (uint32_t k = 0; k < 8; k++) lo[k] = (uint32_t)(w >> (8 * k)) & 0xffu;
  } else {
    uint16_t w[16];
    memcpy(w, __builtin_assume_aligned(static_cast<const uint16_t*>(base) + e, 16), 32);
#pragma unroll
    for (uint32_t k = 0; k < 16; k++) lo[k] = w[k] & 0x3ffu;
  }
}

template <bool BIG> NMX_KERNEL void NMX_LAUNCH_BOUNDS(1024) k_hist_lo(PartBufs b) {
  using Cfg = PartCfg<BIG>;
  NMX_LDS uint32_t tilestart[Cfg::kMaxHi + 1], cnt[Cfg::kMaxLo];
  uint32_t bin, j, first, len;
  if (!tile_of_block<BIG>(b, tilestart, bin, j, first, len)) return;
  const uint32_t t = NMX_TID;
  if (t < Cfg::kMaxLo) cnt[t] = 0;
  NMX_SYNC();
  const uint32_t e0 = t * Cfg::kTilePer;
  if (e0 < len) {
    uint32_t lo[Cfg::kTilePer];
    load_lo<BIG>(b.ent_lo, (size_t)first + e0, lo);
#pragma unroll
    for (uint32_t k = 0; k < Cfg::kTilePer; k++)
      if (e0 + k < len) lds_count(cnt, lo[k]);
  }
  NMX_SYNC();
  if (t < b.ps.nlo && cnt[t]) nmx_atomic_add(&b.bucket_cnt[(bin << b.ps.LB) + t], cnt[t]);
}

template <bool BIG> NMX_KERNEL void NMX_LAUNCH_BOUNDS(1024) k_part_lo(PartBufs b) {
  using Cfg = PartCfg<BIG>;
  using LoT = typename Cfg::LoT;
  NMX_LDS uint32_t stage[Cfg::kTile];
  NMX_LDS LoT stage_lo[Cfg::kTile];
  NMX_LDS uint32_t tilestart[Cfg::kMaxHi + 1], bcnt[Cfg::kMaxLo], bstart[Cfg::kMaxLo + 1], cnt[Cfg::kMaxLo], lbase[Cfg::kMaxLo + 1],
      gbase[Cfg::kMaxLo], wtot[16];
  uint32_t bin, j, first, len;
  if (!tile_of_block<BIG>(b, tilestart, bin, j, first, len)) return;
  const uint32_t t = NMX_TID, bs = NMX_BDIM, nlo = b.ps.nlo, k0 = bin << b.ps.LB;
  if (t < Cfg::kMaxLo) {
    cnt[t] = 0;
    bcnt[t] = (t < nlo && k0 + t < b.nbuckets) ? b.bucket_cnt[k0 + t] : 0;
  }
  NMX_SYNC();
  block_excl_scan(bcnt, bstart, nlo, wtot);  // the bin's buckets inside the bin's region of the final array
  const uint32_t region = b.tab[kTabStride + bin];
  if (j == 0 && t < nlo && k0 + t < b.nbuckets) {  // first tile of the bin: publish [start, end) of its buckets
    b.start[k0 + t] = region + bstart[t];
    b.end[k0 + t] = region + bstart[t + 1];
  }
  uint32_t v[Cfg::kTilePer], lo[Cfg::kTilePer], pos[Cfg::kTilePer];
  const uint32_t e0 = t * Cfg::kTilePer;
  if (e0 < len) {
    load_lo<BIG>(b.ent_lo, (size_t)first + e0, lo);
    memcpy(v, __builtin_assume_aligned(b.ent_val + first + e0, 16), 4 * Cfg::kTilePer);  // aligned 16-byte loads
#pragma unroll
    for (uint32_t k = 0; k < Cfg::kTilePer; k++)
      if (e0 + k < len) pos[k] = lds_rank(cnt, lo[k]);  // arrival rank inside the tile's sub-bin
  }
  NMX_SYNC();
  block_excl_scan(cnt, lbase, nlo, wtot);
  if (t < nlo && cnt[t]) gbase[t] = region + bstart[t] + nmx_atomic_add(&b.bucket_cur[k0 + t], cnt[t]);  // the tile's run
  if (e0 < len) {
#pragma unroll
    for (uint32_t k = 0; k < Cfg::kTilePer; k++)
      if (e0 + k < len) {
        const uint32_t slot = lbase[lo[k]] + pos[k];
        stage[slot] = v[k];
        stage_lo[slot] = (LoT)lo[k];
      }
  }
  NMX_SYNC();
  for (uint32_t sl = t; sl < len; sl += bs) {
    const uint32_t l = stage_lo[sl];
    b.vals[gbase[l] + (sl - lbase[l])] = stage[sl];
  }
}

}  // namespace nmx
