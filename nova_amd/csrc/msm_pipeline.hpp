// msm_pipeline.hpp -- backend-agnostic orchestration of one MSM (see msm_kernels.hpp for the stages).
//
// `BE` supplies memory (a bump arena), kernel launch, pair sort and a blocking device->host copy:
//   DeviceBackend (msm.hip)         HIP stream + rocPRIM radix sort         -> the product path
//   HostEmulBackend (tests/host_emul) plain loops + std::sort, g++ only     -> indexing debug aid in tests
#pragma once
#include <stddef.h>
#include "msm_partition.hpp"
#include "msm_seg.hpp"

namespace nmx {

static constexpr uint32_t kSegMinTotalAuto = 0xfffffffeu;  // MsmArgs::seg_min_total: the default rule of msm_pipeline()
struct MsmArgs {
  const uint32_t* scalars;  // backend-addressable: n x 8 u32 (field) or n x 2 u32 (u64 mode)
  const void* bases;        // backend-addressable AffineW[n]: internal form, canonical
  uint32_t n;
  uint32_t scalars_mont;
  uint32_t u64_bits;   // 0 => field scalars
  uint32_t force_c;    // 0 => heuristic
  uint32_t force_lmax = 0;  // 0 => heuristic (tuning knob: NMX_TUNE_LMAX)
  uint32_t force_fold_t = 0;  // 0 => heuristic (tuning knob: NMX_TUNE_FOLD_T; 1 = no middle fold pass)
  // precomputed-table mode (key registered with NMX_BASES_PRECOMPUTE): `bases` points at T_0[0], tables are
  // pre_stride apart, this call uses entries [pre_offset, pre_offset + n) of each, window width pre_c
  uint32_t pre_stride = 0, pre_offset = 0, pre_c = 0;
  // sparse forms (commit_sparse / commit_sparse_binary): pair i uses base gather[i]; all_ones: scalars are all 1
  const uint32_t* gather = nullptr;
  uint32_t all_ones = 0;
  uint32_t bases_clean = 0;  // 1: the key is known to hold no identity point: the digit stage need not read the bases
  uint32_t no_partition = 0; // 1: generic radix-sort path even where the hand-written partition applies (tests / A-B runs)
  uint32_t seg_min_total = kSegMinTotalAuto;  // segment-balanced accumulate (msm_seg.hpp): automatic, or from this many sorted entries on
  uint32_t seg_min_len = 8;           // shortest segment a lane is given
  uint32_t accum_prefetch = 1;        // gathers in flight ahead of the addition (AccumSegFn PF)
  uint32_t hist_grid = 0;             // blocks of the first-level counting pass (0: as many as the placing pass; tuning)
  uint32_t hist_bs = 0;               // threads per block of the counting pass (0: as the placing pass; it stages nothing, so any multiple of 64 >= the bin count works)
  uint32_t big_slice = 0;             // pieces per block of the big-bucket pass (< 32: SegPlan::big_slice_for)
  uint32_t seg_heavy_above = 0;       // pieces FinalSegFn sums per bucket without a pre-fold (0: SegPlan::heavy_above_for)
  // fused batch over the key's tables (DigitSrc::batch_*): n = sum of the vector lengths, `scalars` unused;
  // wsum_host[j] receives vector j's sum
  uint32_t batch_k = 0;
  const uint32_t* batch_off = nullptr;
  const uint32_t* const* batch_vec = nullptr;
};

inline uint32_t ilog2_u32(uint32_t v) {
  uint32_t r = 0;
  while (v >>= 1) r++;
  return r;
}

// Window width.  Buckets cost ~2 full adds each in the reduction tree, points cost one mixed add per window:
// the optimum grows like log2(n) - O(1); 16 keeps keys in 20 bits and the bucket array in L2-friendly sizes.
// Among c0-1, c0, c0+1 prefer the width whose TOP window is well populated: a top window of t bits sends every
// point into 2^(t-1) buckets, and with t of 2 or 3 those buckets each collect a large share of all points.
inline uint32_t settle_c(int c0, int lo, int hi, uint32_t bits) {
  int best = c0, best_score = -1;
  for (int c = c0 - 1; c <= c0 + 1; c++) {
    if (c < lo || c > hi) continue;
    int W = ((int)bits + 1 + c - 1) / c;
    int top = (int)bits + 1 - (W - 1) * c;           // bits carried by the top window
    int score = (top * 2 >= c ? 1000 : top * 10) - (c == c0 ? 0 : 1);
    if (score > best_score) {
      best_score = score;
      best = c;
    }
  }
  return (uint32_t)best;
}
inline uint32_t choose_c(uint32_t n, uint32_t bits) {
  uint32_t lg = ilog2_u32(n < 2 ? 2 : n);
  int c = (int)lg - 4;
  if (c < 3) c = 3;
  if (c > 16) c = 16;
  if ((uint32_t)c > bits + 1) c = bits + 1;
  if (c >= 6 && bits > 64) c = (int)settle_c(c, 5, 16, bits);
  return (uint32_t)c;
}

// Window width of the precomputed tables of a key of n_key points, from measured sweeps
// (profiles/r01_msm_2p20/window_width_sweep.txt for >= 2^20; profiles/r02_msm_2p20/window_width_sweep_partition.txt for
// the smaller keys, re-swept with the hand-written partition, whose cost no longer steps with the key width):
//   >= 2^22 points  c = 20   13 windows instead of 16: -19 % mixed additions; 2^19 buckets need that many points
//   >= 2^20         c = 17   15 windows instead of 16: -6 % mixed additions for one more reduction level; the 16-bit keys
//                            still take the narrow partition geometry (2^20: 1.71-1.74 ms against 1.77-1.78 at c = 16,
//                            2^21: 2.98 against 3.11; 2^19: a tie -- profiles/r02_msm_2p20/window_width_sweep_final.txt)
//   >= 2^17         c = 16   2^15 buckets (2^17: 0.503 ms against 0.515 at c = 15, 0.683 at c = 8)
//   >= 2^14         c = 15   one level less in the (latency-bound) reduction tree (2^15: 0.373 against 0.410 at c = 8;
//                            2^16: 0.415 against 0.438 at c = 16)
//   below           c = 8    2^7 buckets: seven tree levels; these MSMs are pure latency (2^13: 0.310, flat in c)
inline uint32_t choose_c_precomp(uint32_t n_key, uint32_t bits) {
  uint32_t lg = ilog2_u32(n_key < 2 ? 2 : n_key);
  if (lg >= 22) return 20;
  if (lg >= 20 && bits > 64) return 17;
  const int c = lg >= 17 ? 16 : lg >= 14 ? 15 : 8;
  return settle_c(c, 8, 16, bits);
}

inline uint32_t batch_sets(uint32_t k) {  // bucket sets of a fused batch of k vectors: the next power of two
  uint32_t p = 1;
  while (p < k) p <<= 1;
  return p;
}
inline MsmShape make_shape(uint32_t n, uint32_t bits, uint32_t force_c, uint32_t pre_c = 0, uint32_t batch_k = 0) {
  MsmShape sh;
  sh.n = n;
  sh.c = pre_c ? pre_c : (force_c ? force_c : choose_c(n, bits));
  sh.W = (bits + 1 + sh.c - 1) / sh.c;
  sh.WB = pre_c ? (batch_k ? batch_sets(batch_k) : 1) : sh.W;
  sh.M = 1u << (sh.c - 1);
  sh.nbuckets = sh.WB * sh.M;
  sh.total = n * sh.W;
  if (pre_c) {
    // every bucket collects ~W*n/M points, split into equal tasks: shorter tasks make the accumulate kernel faster
    // (more, shorter waves: smaller tails) and the folds slower.  Measured sweeps 10..64
    // (profiles/r01_msm_2p20/lmax_sweep.txt): 24 is the minimum of accumulate + fold for the c <= 16 tables
    // (512 points per bucket at 2^20), 16 for the c = 20 tables (~100 points per bucket)
    // Below 2^18 pairs the kernel is a handful of waves per SIMD and its time is the length of the task chain: 8
    // (n < 2^15) and 12 (n < 2^18) measured best there (2^13: 0.341 -> 0.312 ms, 2^14: 0.380 -> 0.337, 2^17: 0.555 -> 0.521)
    sh.lmax = n < (1u << 15) ? 8 : n < (1u << 18) ? 12 : pre_c >= 18 ? 16 : 24;
  } else {
    uint32_t avg = n / sh.M;
    sh.lmax = 4 * avg < 32 ? 32 : 4 * avg;
  }
  return sh;
}

// Runs stages 1-7.  On return `wsum_host[0..WB)` holds the per-window sums (XYZZ, Montgomery), and *err_host the
// device error bits.  Returns the shape used.
template <class BE, int FID, int SFID>
MsmShape msm_pipeline(BE& be, const MsmArgs& a, uint32_t scalar_bits, XYZZW* wsum_host,
                      uint32_t* err_host) {
  const uint32_t bits = a.u64_bits ? a.u64_bits : scalar_bits;
  MsmShape sh = make_shape(a.n, bits, a.force_c, a.pre_stride ? a.pre_c : 0, a.pre_stride ? a.batch_k : 0);
  if (a.force_lmax) sh.lmax = a.force_lmax;
  const size_t total = sh.total;
  const uint32_t heavy_cap = (uint32_t)(total / sh.lmax) + 1;
  const uint32_t extra_cap = 2 * heavy_cap;

  const bool part = partition_supported(sh, a.pre_stride != 0) && !a.no_partition;
  const uint32_t seg_lanes = part ? be.template seg_lanes<FID>(total) : 0;
  // Segment-balanced accumulate (msm_seg.hpp) or tasks (AccumFn)?  Default rule (profiles/r03_msm_2p20/seg_threshold.txt): segments
  // when the key has enough buckets for them -- c >= 15 tables (>= 16 384 buckets): 2^14 0.324 ms against 0.341, 2^15 0.362 / 0.373,
  // 2^17 0.490 / 0.510, 2^18 0.659 / 0.678 -- but never on the c = 8 tables of small keys, whose 128 buckets would each span
  // hundreds of lanes (2^13: 0.492 ms against 0.309).  An explicit seg_min_total (option / NMX_TUNE_SEG_MIN_TOTAL) is a plain
  // threshold on the sorted entries.
  const bool seg_auto = a.seg_min_total == kSegMinTotalAuto;
  const bool seg = part && seg_lanes && (seg_auto ? (sh.c >= 15 ? total >= (1u << 17) : total >= (1u << 22)) : total >= a.seg_min_total);
  // Everything that must start as zero lives in ONE block, cleared by ONE fill: bucket bounds, counters, the tickets of the
  // big-bucket pass and the partition's histograms / cursors.  (Round 2 issued five fills per MSM, 3-5 us each plus the
  // gap before the next launch, and cleared the 9 MB bucket_raw array that every reader only touches where it was written.)
  const uint32_t big_cap_s = seg ? (seg_lanes / (SegPlan::kBigAbove + 1) + 1 < sh.nbuckets ? seg_lanes / (SegPlan::kBigAbove + 1) + 1 : sh.nbuckets) : 0;
  const size_t nbounds = 2 * ((size_t)sh.nbuckets + 1) + 8;
  const size_t nctr = part ? 2048 + 3 * kTabStride + 1 + 2 * (size_t)sh.nbuckets : 0;
  const uint32_t big_slice = seg ? SegPlan::big_slice_for(a.big_slice, seg_lanes) : 1;
  const uint32_t big_items_cap = seg ? seg_lanes / big_slice + big_cap_s + 1 : 0;                      // sum of ceil(cnt / slice)
  const uint32_t big_groups_cap = seg ? big_items_cap / SegPlan::kBigGroup + big_cap_s + 1 : 0;       // sum of ceil(slices / 32)
  const size_t nzero = nbounds + big_cap_s + big_groups_cap + nctr;
  uint32_t* start = be.template alloc<uint32_t>(nzero);
  uint32_t* end = start + sh.nbuckets + 1;
  uint32_t* counters = end + sh.nbuckets + 1;  // [0] extra tasks, [1] split buckets, [2] error bits, [3] max tasks, [4] big, [5] non-zero digits, [6] items, [7] groups of the big-bucket pass
  uint32_t* big_done = start + nbounds;
  uint32_t* big_gdone = big_done + big_cap_s;
  uint32_t* ctr = big_gdone + big_groups_cap;
  HeavyRec* heavy = be.template alloc<HeavyRec>(seg ? 1 : heavy_cap);
  const uint32_t big_cap = (uint32_t)(total / ((size_t)64 * sh.lmax)) + 1;
  HeavyRec* big = be.template alloc<HeavyRec>(seg ? 1 : big_cap);
  TaskRec* extra = be.template alloc<TaskRec>(seg ? 1 : extra_cap);
  XYZZW* buckets = be.template alloc<XYZZW>(sh.nbuckets);
  XYZZW* partials = be.template alloc<XYZZW>(seg ? 1 : extra_cap);
  // (the whole 256-byte-padded allocation: a size that is not a multiple of the fill kernel's vector width costs a second dispatch)
  be.memset0(start, (nzero * sizeof(uint32_t) + 255) & ~(size_t)255);

  DigitSrc<SFID> src;
  src.scalars = a.scalars;
  src.bases = a.bases_clean ? nullptr : (const uint32_t*)a.bases;
  src.err = counters + 2;
  src.sh = sh;
  src.scalars_mont = a.scalars_mont;
  src.u64_bits = a.u64_bits;
  src.pre_stride = a.pre_stride;
  src.pre_offset = a.pre_offset;
  src.gather = a.gather;
  src.all_ones = a.all_ones;
  src.batch_k = a.pre_stride ? a.batch_k : 0;
  src.batch_off = a.batch_off;
  src.batch_vec = a.batch_vec;

  uint32_t* vals1;
  if (part) {
    // hand-written two-level LDS partition fused with digit extraction (msm_partition.hpp)
    PartArgs<SFID> pa;
    pa.src = src;
    PartBufs& pb = pa.b;
    pb.ps = make_part_shape(sh, a.pre_stride != 0);
    pb.nbuckets = sh.nbuckets;
    pb.hist_hi = ctr;
    pb.cur_hi = ctr + 1024;
    pb.tab = ctr + 2048;
    pb.bucket_cnt = ctr + 2048 + 3 * kTabStride + 1;
    pb.bucket_cur = pb.bucket_cnt + sh.nbuckets;
    pb.ent_val = be.template alloc<uint32_t>(pb.ps.ent_cap);
    pb.ent_lo = be.template alloc<uint8_t>((size_t)pb.ps.ent_cap * (pb.ps.big ? 2 : 1));
    vals1 = be.template alloc<uint32_t>(total);
    pb.vals = vals1;
    pb.start = start;
    pb.end = end;
    pb.total_out = counters + 5;
    // the widths the tables are built with get their own instantiation (constant bit positions); BIG = keys wider than 15 bits
    pb.single_bin = pb.ps.nhi == 1 ? 1u : 0u;
    auto level1 = [&](auto hist, auto part) {
      if (pb.single_bin) {  // (PartBufs::single_bin) placing pass first, the tile table from its cursor, no counting pass
        be.mark("digits");
        be.mark("sort");
        be.launch_kernel(part, pb.ps.grid1, pb.ps.bs1, pa);
        PartBufs pt = pb;
        pt.hist_hi = pb.cur_hi;
        if (pb.ps.big) be.launch_kernel(&k_tiles<true>, 1u, 1024u, pt);
        else be.launch_kernel(&k_tiles<false>, 1u, 1024u, pt);
        return;
      }
      be.mark("digits");
      // the counting pass ends with one global atomic per block and bin: fewer, longer-lived blocks (a.hist_grid) contend less
      {
        uint32_t hbs = pb.ps.bs1, hgrid = pb.ps.grid1;
        if (a.hist_bs >= 64 && a.hist_bs <= 1024 && (a.hist_bs & 63u) == 0) {
          hbs = a.hist_bs;
          const uint32_t chunks = (sh.n + hbs - 1) / hbs;
          hgrid = chunks < 4096 ? chunks : 4096;
        }
        if (a.hist_grid && a.hist_grid < hgrid) hgrid = a.hist_grid;
        be.launch_kernel(hist, hgrid, hbs, pa);
      }
      if (pb.ps.big) be.launch_kernel(&k_tiles<true>, 1u, 1024u, pb);
      else be.launch_kernel(&k_tiles<false>, 1u, 1024u, pb);
      be.mark("sort");
      be.launch_kernel(part, pb.ps.grid1, pb.ps.bs1, pa);
    };
    if (pb.ps.big) {  // c = 20 tables, or the bucket sets of a fused batch over c = 15 / 16 / 17 tables
      if (sh.c == 20) level1(&k_hist_hi<SFID, 20, true>, &k_part_hi<SFID, 20, true>);
      else if (sh.c == 17) level1(&k_hist_hi<SFID, 17, true>, &k_part_hi<SFID, 17, true>);
      else if (sh.c == 16) level1(&k_hist_hi<SFID, 16, true>, &k_part_hi<SFID, 16, true>);
      else if (sh.c == 15) level1(&k_hist_hi<SFID, 15, true>, &k_part_hi<SFID, 15, true>);
      else level1(&k_hist_hi<SFID, 0, true>, &k_part_hi<SFID, 0, true>);
      be.launch_kernel(&k_hist_lo<true>, pb.ps.tiles_cap, kTileThreads, pb);
      be.launch_kernel(&k_part_lo<true>, pb.ps.tiles_cap, kTileThreads, pb);
    } else {
      switch (sh.c) {
        case 17: level1(&k_hist_hi<SFID, 17, false>, &k_part_hi<SFID, 17, false>); break;
        case 16: level1(&k_hist_hi<SFID, 16, false>, &k_part_hi<SFID, 16, false>); break;
        case 15: level1(&k_hist_hi<SFID, 15, false>, &k_part_hi<SFID, 15, false>); break;
        case 8: level1(&k_hist_hi<SFID, 8, false>, &k_part_hi<SFID, 8, false>); break;
        default: level1(&k_hist_hi<SFID, 0, false>, &k_part_hi<SFID, 0, false>);
      }
      be.launch_kernel(&k_hist_lo<false>, pb.ps.tiles_cap, kTileThreads, pb);
      be.launch_kernel(&k_part_lo<false>, pb.ps.tiles_cap, kTileThreads, pb);
    }
    be.mark("bounds");
  } else {
    uint32_t* keys0 = be.template alloc<uint32_t>(total);
    uint32_t* vals0 = be.template alloc<uint32_t>(total);
    uint32_t* keys1 = be.template alloc<uint32_t>(total);
    vals1 = be.template alloc<uint32_t>(total);
    be.mark("digits");
    {
      DigitsFn<SFID> f{src, keys0, vals0};
      be.launch(f, sh.n);
    }
    be.mark("sort");
    {
      uint32_t key_bits = ilog2_u32(sh.nbuckets) + 1;  // keys in [0, nbuckets]
      be.sort_pairs(keys0, keys1, vals0, vals1, total, key_bits);
    }
    be.mark("bounds");
    {
      BoundsFn f{keys1, start, end, (uint32_t)total};
      be.launch(f, (uint32_t)((total + BoundsFn::kPerLane - 1) / BoundsFn::kPerLane));
    }
  }
  if (seg) {
    // large MSM: equal segments of the entry array per resident lane, raw-limb pieces, short folds (msm_seg.hpp)
    XYZZL* partial_raw = be.template alloc<XYZZL>(seg_lanes);
    // a listed bucket spans more than 8 (64) segments: at most seg_lanes / 9 (/ 65) of them, whatever the distribution
    auto list_cap = [&](uint32_t above) {
      const uint32_t by_lanes = seg_lanes / (above + 1) + 1;
      return by_lanes < sh.nbuckets ? by_lanes : sh.nbuckets;
    };
    const uint32_t heavy_above = a.seg_heavy_above ? a.seg_heavy_above : SegPlan::heavy_above_for(seg_lanes, sh.nbuckets);
    HeavyRec* heavy_s = be.template alloc<HeavyRec>(list_cap(heavy_above));
    HeavyRec* big_s = be.template alloc<HeavyRec>(big_cap_s);
    // not cleared: every non-empty bucket's first piece is written by the lane its first entry falls into, and the passes
    // below read bucket_raw[k] for non-empty buckets only
    XYZZL* bucket_raw = be.template alloc<XYZZL>(sh.nbuckets);
    const uint32_t* total_p = counters + 5;
    uint32_t* big_items = be.template alloc<uint32_t>(big_items_cap);
    uint32_t* big_gbase = be.template alloc<uint32_t>(big_cap_s);
    const SegPlan plan{start, end, total_p, counters, heavy_s, big_s, sh.nbuckets, seg_lanes, a.seg_min_len, heavy_above, big_items, big_gbase, big_slice};
    be.mark("accum");
    if (a.accum_prefetch > 1) {
      AccumSegFn<FID, 2> f{(const AffineW*)a.bases, vals1, start, end, total_p, bucket_raw, partial_raw, sh.nbuckets,
                           seg_lanes, a.seg_min_len, plan};
      be.launch(f, seg_lanes);
    } else {
      AccumSegFn<FID, 1> f{(const AffineW*)a.bases, vals1, start, end, total_p, bucket_raw, partial_raw, sh.nbuckets,
                           seg_lanes, a.seg_min_len, plan};
      be.launch(f, seg_lanes);
    }
    be.mark("fold");
    // big buckets (> 64 pieces) completely; then the heavy ones (> heavy_above) down to heavy_above positions; then
    // every other bucket.  On uniformly random scalars the first two launches find empty lists and exit.
    be.template launch_big_all<FID>(counters, big_s, big_items, big_gbase, bucket_raw, partial_raw, buckets, big_done, big_gdone, big_slice);
    if (heavy_above < SegPlan::kBigAbove) {
      // listed buckets number at most seg_lanes / (heavy_above + 1); when the typical bucket is not heavy (c = 17 tables:
      // 9 pieces against 12) a small grid walks whatever the input made heavy
      const bool expected = seg_lanes / sh.nbuckets > heavy_above;
      uint32_t groups = list_cap(heavy_above);
      if (!expected && groups > 2048) groups = 2048;
      be.template launch_fold_raw<FID>(counters, heavy_s, partial_raw, heavy_above, 0xffffffffu, groups, 0u);
    }
    be.template launch_final_seg<FID>(start, end, total_p, bucket_raw, partial_raw, buckets, sh.nbuckets, seg_lanes,
                                      a.seg_min_len, heavy_above);
  } else if (sh.nbuckets <= 1024 && be.small_chunk() != 0) {
    // small MSM (c = 8 tables of keys below 2^14 points, plain keys of a few hundred pairs): two block-level launches
    // instead of plan + expand + accumulate + six strided folds (curve_quad.hpp k_small_accum)
    const uint32_t chunk = be.small_chunk();
    const uint32_t blocks = (uint32_t)(total / chunk) + sh.nbuckets + 1;  // sum of ceil(bucket / chunk) stays below this
    XYZZW* part_s = be.template alloc<XYZZW>((size_t)blocks * 64);  // one partial per quad
    be.mark("accum");
    be.template launch_small_accum<FID>((const AffineW*)a.bases, vals1, start, end, part_s, buckets, sh.nbuckets, chunk, blocks);
    be.mark("fold");
    be.template launch_small_combine<FID>((const AffineW*)a.bases, vals1, start, end, part_s, buckets, sh.nbuckets, chunk);
  } else {
  {
    PlanFn f{start, end, counters, heavy, big, sh};
    be.launch(f, (sh.nbuckets + PlanFn::kPerLane - 1) / PlanFn::kPerLane);  // each lane plans kPerLane buckets
    const uint32_t hb = heavy_cap < sh.nbuckets ? heavy_cap : sh.nbuckets;
    ExpandFn e{start, end, counters, heavy, extra, sh, 64, hb < 16384 ? hb : 16384};
    be.launch(e, e.groups * e.lanes);
  }
  be.mark("accum");
  {
    be.template launch_accum<FID>((const AffineW*)a.bases, vals1, start, end, counters, extra, buckets, partials, sh,
                                  sh.nbuckets + extra_cap, (uint64_t)sh.nbuckets + total / sh.lmax);
  }
  be.mark("fold");
  {
    // Every pass shrinks a bucket's partial count by at most 8x (<= 7 dependent additions per lane); passes that no
    // bucket needs exit at once (max task count is on the device).  One bucket holding a whole window (all-equal or
    // 0/1 scalars) therefore costs ~35 dependent additions instead of N / (lmax * 256).
    const uint32_t hb = heavy_cap < sh.nbuckets ? heavy_cap : sh.nbuckets;  // upper bound on heavy buckets
    // typical task count per bucket decides the last split: ~22 partials fold fastest as 4 lanes x 6 then 1 x 4
    const uint32_t typical = (uint32_t)(total / ((size_t)sh.nbuckets * sh.lmax));
    uint32_t mid = typical > 32 ? 8u : 4u;
    // tuning: 1 = no middle pass; otherwise 2..63 lanes (a wider middle pass would overlap the 64-lane pass before it)
    if (a.force_fold_t) mid = a.force_fold_t == 1 ? 64 : (a.force_fold_t > 63 ? 63 : a.force_fold_t);
    const uint32_t Ts[6] = {32768, 4096, 512, 64, mid, 1};
    for (int p = 0; p < 6; p++) {
      const uint32_t T = Ts[p];
      const uint32_t cap = p == 0 ? 0xffffffffu : Ts[p - 1];
      if (p == 4 && T == 64) continue;
      if ((uint64_t)T * sh.lmax > total && T != 1) continue;  // no bucket can have more than T tasks
      // buckets with more than T tasks number at most total / (T * lmax): ~2^18 lanes per pass cover them in a few
      // sweeps, and a pass nobody needs costs one small empty launch
      // T >= 64 passes walk the short list of big buckets; the last two walk every split bucket, one group each
      const uint32_t bound = T >= 64 ? (big_cap < hb ? big_cap : hb) : hb;
      uint32_t groups = T >= 64 ? (1u << 18) / T : bound;
      if (groups > bound) groups = bound;
      if (groups < 1) groups = 1;
      be.template launch_fold<FID>(counters, T >= 64 ? big : heavy, partials, buckets, T, cap, groups);
    }
  }
  }
  be.mark("reduce");
  bool err_appended = false;  // the last tree launch parks the error word behind the sums: one copy instead of two
  const XYZZW* Y = be.template reduce_tree<FID>(buckets, sh, counters + 2, &err_appended);
  be.mark("tail");
  if (err_appended) {
    be.d2h_split(wsum_host, sizeof(XYZZW) * sh.WB, err_host, sizeof(uint32_t), Y);
  } else {
    be.d2h(wsum_host, Y, sizeof(XYZZW) * sh.WB);
    be.d2h(err_host, counters + 2, sizeof(uint32_t));
  }
  be.mark("end");
  be.sync();
  return sh;
}

// Host tail: Horner over window sums, high to low (msm.rs:651-661).
template <int FID> XYZZ<FID> combine_windows(const XYZZW* wsum, const MsmShape& sh) {
  XYZZ<FID> acc = XYZZ<FID>::load(wsum[sh.WB - 1]);
  for (int w = (int)sh.WB - 2; w >= 0; w--) {  // WB == 1 (precomputed tables): nothing to combine
    for (uint32_t q = 0; q < sh.c; q++) acc.dbl_in_place();
    acc.add(XYZZ<FID>::load(wsum[w]));
  }
  return acc;
}

}  // namespace nmx
