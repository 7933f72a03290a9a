// msm_seg.hpp -- segment-balanced bucket accumulation (large MSMs on the table path).
//
// Round 1's accumulate kernel gave every lane one TASK (<= lmax = 24 consecutive entries of one bucket): ~730 k tasks
// at 2^20 = 3.7 rounds of the chip's resident lanes, so the last round ran partly empty (7 % of the kernel), and
// every bucket was left as ~22 partial sums for the fold passes (0.13 ms).  Here the sorted entry array [0, total) is
// cut into exactly as many equal SEGMENTS as lanes are resident (CUs x waves x 64), whatever the bucket boundaries:
//
//   AccumSegFn   lane L adds entries [L * seg, (L + 1) * seg) in order, crossing bucket boundaries as they come.  The
//                piece of a bucket that STARTS inside the lane is written to bucket_raw[k]; the piece that continues a
//                bucket begun in an earlier lane (at most one per lane, its first) to partial_raw[L].  Hence
//                     bucket k = bucket_raw[k] + sum of partial_raw[L] for L in (L0, L1],  L0 = start[k] / seg,
//                                                                                             L1 = (end[k] - 1) / seg
//                -- a contiguous range.  All lanes do the same number of mixed additions (+-1): no tail round.
//   SegPlan      (run by the first lanes of AccumSegFn itself: start / end are final before it starts) buckets spanning
//                more than heavy_above lanes -> `heavy` list, more than 64 -> `big` list
//   BigBucketFn  every big bucket completely: one block per 4096-piece slice, strided sums + an LDS tree; a bucket of
//                several slices is finished by the block that arrives last (ticket counter, no spinning) -- ONE launch
//                for every bucket size up to a whole window in one bucket (all-equal scalars); curve_quad.hpp k_big_all
//   FoldRawFn    one strided pre-fold (T = heavy_above) of the heavy ranges
//   FinalSegFn   every other bucket: bucket_raw[k] + its (<= heavy_above remaining) partials -> canonical XYZZW
// Round 2 ran Plan + five FoldRaw passes (four of them over `big`, empty on ordinary inputs) + Final: seven dependent
// launches, ~55 us of them doing nothing; now three, of which two exit at once on uniformly random scalars.
//
// A bucket of 512 entries spans ~6 segments of 86: the fold work per bucket drops from ~22 partials to ~6.
// Pieces are stored as raw 9-limb coordinates (XYZZL, 144 B): a flush sits inside a divergent branch (lanes of a wave
// cross bucket boundaries at different entries), so it must be a handful of stores, not four canonicalisations.
#pragma once
#include "msm_kernels.hpp"

namespace nmx {

// entries per lane for `total` entries over `lanes` lanes (computed on the device: `total` is only known there)
NMX_HD uint32_t seg_len(uint32_t total, uint32_t lanes, uint32_t min_seg) {
  const uint32_t s = (uint32_t)(((uint64_t)total + lanes - 1) / lanes);
  return s < min_seg ? min_seg : s;
}

// PF = how many gathers are in flight ahead of the addition being computed (1: as AccumFn; 2: one more 64-byte row
// in registers, for the case where the gather latency under full load exceeds one addition of the wave's neighbours)
// Buckets whose entries span many lanes: lists for the pre-fold / big-bucket passes.  counters: [1] heavy count,
// [3] largest partial count among the big buckets, [4] big count.  Wave-safe on the device (one atomic per wave).
struct SegPlan {
  static constexpr uint32_t kBigAbove = 64;
  // The big-bucket pass works on slices of a bucket's pieces, one block per slice: every (bucket, slice) is an ITEM, numbered
  // here (counters[6] = items so far; rec.pad = the bucket's first item; big_items[item] = its position in the big list), so that
  // the pass is a plain strided loop over items whatever mix of bucket sizes the input produced.  Slices are summed in groups of
  // kBigGroup (counters[7] = groups so far, big_gbase[h] = the bucket's first group): a bucket of 768 slices ends with 24 group
  // sums, not with one block adding 768 points.
  static constexpr uint32_t kBigGroup = 32;
  // slice length: ~768 items when every lane's piece belongs to a big bucket (the pass keeps 1024 blocks resident)
  static uint32_t big_slice_for(uint32_t forced, uint32_t lanes) {
    if (forced >= 32) return forced;
    const uint32_t s = (lanes / 768 + 31u) & ~31u;
    return s < 64 ? 64 : s;
  }
  // A bucket with more than `heavy_above` continuation pieces is listed for the T = heavy_above pre-fold; FinalSegFn sums up
  // to that many serially.  lanes / nbuckets pieces per bucket on uniformly random scalars: 18 at c = 16 (8: one pre-fold
  // pass halves them), 9 at c = 17 (12: no bucket is listed, no pre-fold work, FinalSegFn adds one more piece).
  static uint32_t heavy_above_for(uint32_t lanes, uint32_t nbuckets) { return lanes / nbuckets <= 12u ? 12u : 8u; }
  const uint32_t* start;
  const uint32_t* end;
  const uint32_t* total_p;
  uint32_t* counters;
  HeavyRec* heavy;  // heavy_above < pieces <= kBigAbove
  HeavyRec* big;    // pieces > kBigAbove
  uint32_t nbuckets, lanes, min_seg, heavy_above;
  uint32_t* big_items;  // [lanes / big_slice + big capacity + 1]
  uint32_t* big_gbase;  // [big capacity]
  uint32_t big_slice;
  // every lane of the wave calls this together (lanes without a bucket: valid = false)
  NMX_HD void operator()(uint32_t k, bool valid) const {
    const uint32_t seg = seg_len(*total_p, lanes, min_seg);
    uint32_t cnt = 0, off = 0;
    if (valid && k < nbuckets) {
      const uint32_t s0 = start[k], e0 = end[k];
      if (e0 > s0) {
        const uint32_t l0 = s0 / seg, l1 = (e0 - 1) / seg;
        cnt = l1 - l0;
        off = l0 + 1;
      }
    }
    const bool is_heavy = cnt > heavy_above && cnt <= kBigAbove, is_big = cnt > kBigAbove;
    uint32_t slot = 0;
#if defined(__HIP_DEVICE_COMPILE__)
    const unsigned long long m = __ballot(is_heavy);
    if (m != 0) {  // wave-uniform
      const uint32_t lane = __lane_id(), leader = (uint32_t)__ffsll((long long)m) - 1u;
      uint32_t base = 0;
      if (lane == leader) base = atomicAdd(&counters[1], (uint32_t)__popcll(m));  // one atomic per wave
      base = __shfl(base, (int)leader);
      slot = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    }
#else
    if (is_heavy) slot = counters[1]++;
#endif
    if (is_heavy) heavy[slot] = HeavyRec{k, off, cnt, 0};
    if (is_big) {  // rare: a bucket holding more than 64 lanes' worth of entries
      nmx_atomic_max(&counters[3], cnt);
      const uint32_t nsl = (cnt + big_slice - 1) / big_slice;
      const uint32_t first_item = nmx_atomic_add(&counters[6], nsl), h = nmx_atomic_add(&counters[4], 1);
      big[h] = HeavyRec{k, off, cnt, first_item};
      big_gbase[h] = nmx_atomic_add(&counters[7], (nsl + kBigGroup - 1) / kBigGroup);
      for (uint32_t t = 0; t < nsl; t++) big_items[first_item + t] = h;
    }
  }
};

template <int FID, int PF = 1> struct AccumSegFn {
  const AffineW* bases;
  const uint32_t* vals;
  const uint32_t* start;
  const uint32_t* end;
  const uint32_t* total_p;  // number of entries (device memory: written by the partition's scan kernel)
  XYZZL* bucket_raw;        // [nbuckets], zero-initialised (= identity)
  XYZZL* partial_raw;       // [lanes]
  uint32_t nbuckets, lanes, min_seg;
  SegPlan plan;  // the bucket lists of the passes after this kernel, written by its first lanes

  // the smallest k >= k0 with end[k] > j (end[] is nondecreasing and end[nbuckets - 1] = total > j)
  NMX_HD uint32_t first_bucket_after(uint32_t k0, uint32_t j) const {
    uint32_t lo = k0, hi = nbuckets - 1;
    while (lo < hi) {
      const uint32_t mid = (lo + hi) >> 1;
      if (end[mid] > j) hi = mid;
      else lo = mid + 1;
    }
    return lo;
  }
  NMX_HD void operator()(uint32_t L) const {
    // plan first: lane L classifies buckets L, L + lanes, ... (the launch has whole waves: `lanes` is a multiple of 256)
    for (uint32_t kb = L & ~63u; kb < nbuckets; kb += lanes) plan(kb + (L & 63u), true);
    const uint32_t total = *total_p;
    const uint32_t seg = seg_len(total, lanes, min_seg);
    const uint64_t a64 = (uint64_t)L * seg;
    if (a64 >= total) return;
    const uint32_t a = (uint32_t)a64, b = (a64 + seg < total) ? a + seg : total;
    uint32_t k = first_bucket_after(0, a), e = end[k];
    // The boundary after next is loaded one bucket ahead: a load issued inside the crossing branch would make the wave
    // wait for everything in flight (vmcnt retires in order: the prefetched gathers, the flush's stores) on EVERY
    // crossing of ANY of its lanes -- ~75 times per 86 additions.
    uint32_t e_next = end[k + 1];  // end[] has nbuckets + 1 entries
    bool head = start[k] < a;      // this lane's first piece continues a bucket begun earlier
    XYZZ<FID> acc = XYZZ<FID>::identity();
    // same software pipeline as AccumFn: indices PF + 1 entries ahead, gathers PF ahead
    uint32_t v = vals[a];
    uint32_t vn = a + 1 < b ? vals[a + 1] : v;
    uint32_t vn2 = (PF > 1 && a + 2 < b) ? vals[a + 2] : vn;
    AffineW cur = bases[v & 0x7fffffffu];
    AffineW nx1 = cur;
    if (PF > 1 && a + 1 < b) nx1 = bases[vn & 0x7fffffffu];
    for (uint32_t j = a; j < b; j++) {
      uint32_t vnn = vn, vnn2 = vn2;
      AffineW nxt = cur, nxt2 = nx1;
      if constexpr (PF == 1) {
        if (j + 1 < b) nxt = bases[vn & 0x7fffffffu];
        if (j + 2 < b) vnn = vals[j + 2];
      } else {
        nxt = nx1;                                          // gathered during the previous addition
        if (j + 2 < b) nxt2 = bases[vn2 & 0x7fffffffu];     // the row for entry j + 2
        vnn = vn2;
        if (j + 3 < b) vnn2 = vals[j + 3];
      }
      if (j == e) {  // bucket k is complete inside this lane (or its continued piece is)
        if (head) acc.store_raw(partial_raw[L]);
        else acc.store_raw(bucket_raw[k]);
        head = false;
        acc = XYZZ<FID>::identity();
        k++;
        e = e_next;
        if (e <= j) {  // empty buckets follow (skewed scalars): search instead of walking them one dependent load at a time
          k = first_bucket_after(k, j);
          e = end[k];
        }
        e_next = end[k + 1];
      }
      acc.add_affine(Affine<FID>::load(cur), (v >> 31) != 0);
      cur = nxt;
      v = vn;
      vn = vnn;
      if constexpr (PF > 1) {
        nx1 = nxt2;
        vn2 = vnn2;
      }
    }
    if (head) acc.store_raw(partial_raw[L]);
    else acc.store_raw(bucket_raw[k]);
  }
};

// FoldFn on raw partials: lane j of group g folds positions j, j + T, j + 2T, ... < min(cnt, cap) of a listed bucket
// into position j (in place).  One pass with T = heavy_above over `heavy` (use_big = 0; use_big = 1 walks the big list).
template <int FID> struct FoldRawFn {
  const uint32_t* counters;
  const HeavyRec* list;
  XYZZL* partial_raw;
  uint32_t T, cap, groups;
  uint32_t use_big;  // 1: walk the big list (counters[4], skip the pass when counters[3] <= T)
  NMX_HD void operator()(uint32_t tid) const {
    if (use_big && counters[3] <= T) return;
    const uint32_t j = tid % T, nh = counters[use_big ? 4 : 1];
    for (uint32_t h = tid / T; h < nh; h += groups) {
      const HeavyRec r = list[h];
      const uint32_t cnt = r.cnt < cap ? r.cnt : cap;
      if (j + T >= cnt) continue;  // nothing to add into position j
      XYZZ<FID> acc = XYZZ<FID>::load_raw(partial_raw[r.off + j]);
      for (uint32_t q = j + T; q < cnt; q += T) acc.template add<kLatTail>(XYZZ<FID>::load_raw(partial_raw[r.off + q]));
      acc.store_raw(partial_raw[r.off + j]);
    }
  }
};

// Every bucket: the piece that started it + the partials that continue it (after the pre-folds at most heavy_above of
// them hold everything) -> canonical XYZZW.
template <int FID> struct FinalSegFn {
  const uint32_t* start;
  const uint32_t* end;
  const uint32_t* total_p;
  const XYZZL* bucket_raw;
  const XYZZL* partial_raw;
  XYZZW* buckets;
  uint32_t nbuckets, lanes, min_seg, heavy_above;
  NMX_HD void operator()(uint32_t k) const {
    const uint32_t s0 = start[k], e0 = end[k];
    XYZZ<FID> acc = XYZZ<FID>::identity();
    if (e0 > s0) {
      const uint32_t seg = seg_len(*total_p, lanes, min_seg);
      const uint32_t l0 = s0 / seg, l1 = (e0 - 1) / seg;
      uint32_t cnt = l1 - l0;
      if (cnt > SegPlan::kBigAbove) return;  // written by the big-bucket pass
      if (cnt > heavy_above) cnt = heavy_above;
      acc = XYZZ<FID>::load_raw(bucket_raw[k]);
      for (uint32_t j = 0; j < cnt; j++) acc.template add<kLatTail>(XYZZ<FID>::load_raw(partial_raw[l0 + 1 + j]));
    }
    acc.store(buckets[k]);
  }
};

// A big bucket completely, one lane: the reference semantics of curve_quad.hpp's k_big_all (which the device runs); used as
// is by the host emulation.
template <int FID> struct BigBucketFn {
  const uint32_t* counters;
  const HeavyRec* big;
  const XYZZL* bucket_raw;
  const XYZZL* partial_raw;
  XYZZW* buckets;
  NMX_HD void operator()(uint32_t h) const {
    if (h >= counters[4]) return;
    const HeavyRec r = big[h];
    XYZZ<FID> acc = XYZZ<FID>::load_raw(bucket_raw[r.bucket]);
    for (uint32_t j = 0; j < r.cnt; j++) acc.add(XYZZ<FID>::load_raw(partial_raw[r.off + j]));
    acc.store(buckets[r.bucket]);
  }
};

}  // namespace nmx
