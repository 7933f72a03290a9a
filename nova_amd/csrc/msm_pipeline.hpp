// msm_pipeline.hpp -- backend-agnostic orchestration of one MSM (see msm_kernels.hpp for the stages).
//
// `BE` supplies memory (a bump arena), kernel launch, pair sort and a blocking device->host copy:
//   DeviceBackend (msm.hip)         HIP stream + rocPRIM radix sort         -> the product path
//   HostEmulBackend (tests/host_emul) plain loops + std::sort, g++ only     -> indexing debug aid in tests
#pragma once
#include <stddef.h>
#include "msm_kernels.hpp"

namespace nmx {

struct MsmArgs {
  const uint32_t* scalars;  // backend-addressable: n x 8 u32 (field) or n x 2 u32 (u64 mode)
  const void* bases;        // backend-addressable AffineW[n]: internal form, canonical
  uint32_t n;
  uint32_t scalars_mont;
  uint32_t u64_bits;   // 0 => field scalars
  uint32_t force_c;    // 0 => heuristic
};

inline uint32_t ilog2_u32(uint32_t v) {
  uint32_t r = 0;
  while (v >>= 1) r++;
  return r;
}

// Window width.  Buckets cost ~2 full adds each in the reduction tree, points cost one mixed add per window:
// the optimum grows like log2(n) - O(1); 16 keeps keys in 20 bits and the bucket array in L2-friendly sizes.
inline uint32_t choose_c(uint32_t n, uint32_t bits) {
  uint32_t lg = ilog2_u32(n < 2 ? 2 : n);
  int c = (int)lg - 4;
  if (c < 3) c = 3;
  if (c > 16) c = 16;
  if ((uint32_t)c > bits + 1) c = bits + 1;
  return (uint32_t)c;
}

inline MsmShape make_shape(uint32_t n, uint32_t bits, uint32_t force_c) {
  MsmShape sh;
  sh.n = n;
  sh.c = force_c ? force_c : choose_c(n, bits);
  sh.W = (bits + 1 + sh.c - 1) / sh.c;
  sh.M = 1u << (sh.c - 1);
  sh.nbuckets = sh.W * sh.M;
  uint32_t avg = n / sh.M;
  sh.lmax = 4 * avg < 32 ? 32 : 4 * avg;
  sh.total = n * sh.W;
  return sh;
}

static constexpr uint32_t kFoldGroups = 512;

// Runs stages 1-7.  On return `wsum_host[0..W)` holds the per-window sums (XYZZ, Montgomery), and *err_host the
// device error bits.  Returns the shape used.
template <class BE, int FID, int SFID>
MsmShape msm_pipeline(BE& be, const MsmArgs& a, uint32_t scalar_bits, XYZZW* wsum_host,
                      uint32_t* err_host) {
  const uint32_t bits = a.u64_bits ? a.u64_bits : scalar_bits;
  MsmShape sh = make_shape(a.n, bits, a.force_c);
  const size_t total = sh.total;
  const uint32_t heavy_cap = (uint32_t)(total / sh.lmax) + 1;
  const uint32_t extra_cap = 2 * heavy_cap;

  uint32_t* keys0 = be.template alloc<uint32_t>(total);
  uint32_t* vals0 = be.template alloc<uint32_t>(total);
  uint32_t* keys1 = be.template alloc<uint32_t>(total);
  uint32_t* vals1 = be.template alloc<uint32_t>(total);
  uint32_t* start = be.template alloc<uint32_t>(2 * ((size_t)sh.nbuckets + 1) + 4);
  uint32_t* end = start + sh.nbuckets + 1;
  uint32_t* counters = end + sh.nbuckets + 1;  // [0] extra tasks, [1] heavy buckets, [2] error bits
  HeavyRec* heavy = be.template alloc<HeavyRec>(heavy_cap);
  TaskRec* extra = be.template alloc<TaskRec>(extra_cap);
  XYZZW* buckets = be.template alloc<XYZZW>(sh.nbuckets);
  XYZZW* partials = be.template alloc<XYZZW>(extra_cap);

  be.memset0(start, (2 * ((size_t)sh.nbuckets + 1) + 4) * sizeof(uint32_t));

  be.mark("digits");
  {
    DigitsFn<SFID> f;
    f.scalars = a.scalars;
    f.bases = (const uint32_t*)a.bases;
    f.keys = keys0;
    f.vals = vals0;
    f.err = counters + 2;
    f.sh = sh;
    f.scalars_mont = a.scalars_mont;
    f.u64_bits = a.u64_bits;
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
    be.launch(f, (uint32_t)total);
  }
  {
    PlanFn f{start, end, counters, heavy, extra, sh};
    be.launch(f, sh.nbuckets);
  }
  be.mark("accum");
  {
    AccumFn<FID> f{(const AffineW*)a.bases, vals1, start, end, counters, extra, buckets, partials, sh};
    be.launch(f, sh.nbuckets + extra_cap);
  }
  be.mark("fold");
  {
    FoldFn<FID> f{counters, heavy, partials, buckets, 256, 0xffffffffu, kFoldGroups};
    be.launch(f, kFoldGroups * 256);
    f.T = 16;
    f.cap = 256;
    be.launch(f, kFoldGroups * 16);
    f.T = 1;
    f.cap = 16;
    be.launch(f, kFoldGroups);
  }
  be.mark("reduce");
  const XYZZW* A = buckets;
  const XYZZW* Y = buckets;
  uint32_t n_in = sh.M, ls = 0, first = 1;
  if (n_in == 1) {
    // c == 1: one bucket per window, weight 1: the bucket is the window sum
  }
  while (n_in > 1) {
    uint32_t m = n_in < 16 ? n_in : 16;
    uint32_t n_out = n_in / m;
    XYZZW* Ao = be.template alloc<XYZZW>((size_t)sh.W * n_out);
    XYZZW* Yo = be.template alloc<XYZZW>((size_t)sh.W * n_out);
    ReduceFn<FID> f{A, Y, Ao, Yo, n_in, m, ls, first};
    be.launch(f, sh.W * n_out);
    A = Ao;
    Y = Yo;
    ls += ilog2_u32(m);
    n_in = n_out;
    first = 0;
  }
  be.mark("tail");
  be.d2h(wsum_host, Y, sizeof(XYZZW) * sh.W);
  be.d2h(err_host, counters + 2, sizeof(uint32_t));
  be.mark("end");
  be.sync();
  return sh;
}

// Host tail: Horner over window sums, high to low (msm.rs:651-661).
template <int FID> XYZZ<FID> combine_windows(const XYZZW* wsum, const MsmShape& sh) {
  XYZZ<FID> acc = XYZZ<FID>::load(wsum[sh.W - 1]);
  for (int w = (int)sh.W - 2; w >= 0; w--) {
    for (uint32_t q = 0; q < sh.c; q++) acc.dbl_in_place();
    acc.add(XYZZ<FID>::load(wsum[w]));
  }
  return acc;
}

}  // namespace nmx
