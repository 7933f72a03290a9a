// tests/host_emul/emul.cpp -- TEST-ONLY serial emulation of the HIP kernels (g++; never linked into the product).
//
// Runs the *same* functor bodies and the same msm_pipeline() orchestration as libnova_mi355x.so, with a backend
// made of plain loops and std::stable_sort.  Purpose: catch indexing / recoding / edge-case bugs in the kernel
// bodies on machines without a GPU (the build container), so GPU minutes are spent on parity and timing rather
// than on debugging.  It is not an oracle (it shares code with the product) and not a fallback (the product
// returns NMX_E_NO_DEVICE without a GPU); tests compare it against oracle/ like any other implementation.
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <numeric>
#include <vector>

#include "simt.hpp"

#include "../../nova_amd/csrc/curves.hpp"
#include "../../nova_amd/csrc/msm_pipeline.hpp"

using namespace nmx;

static unsigned long long g_block_kernel_launches = 0;
static uint32_t g_seg_lanes = 37;             // emul_set_seg_lanes
static uint32_t g_seg_min_total = 1u << 21;  // emul_set_seg_min_total: lower it to route small MSMs through msm_seg.hpp

struct HostEmulBackend {
  std::vector<void*> blocks;
  ~HostEmulBackend() {
    for (void* p : blocks) free(p);
  }
  template <class T> T* alloc(size_t count) {
    void* p = calloc(((count ? count : 1) * sizeof(T) + 255) & ~(size_t)255, 1);  // padded to 256 bytes like the device arena (the pipeline clears whole allocations)
    blocks.push_back(p);
    return (T*)p;
  }
  void memset0(void* p, size_t bytes) { memset(p, 0, bytes); }
  template <class F> void launch(const F& f, uint32_t n) {
    for (uint32_t t = 0; t < n; t++) f(t);
  }
  template <class A> void launch_kernel(void (*k)(A), uint32_t grid, uint32_t block, const A& a) {
    if (grid == 0) return;
    g_block_kernel_launches++;
    simt::launch(grid, block, [&] { k(a); });  // one fiber per thread, real barriers (tests/host_emul/simt.hpp)
  }
  template <int FID>
  void launch_accum(const AffineW* bases, const uint32_t* vals, const uint32_t* start, const uint32_t* end,
                    const uint32_t* counters, const TaskRec* extra, XYZZW* buckets, XYZZW* partials, const MsmShape& sh,
                    uint32_t slots, uint64_t) {
    AccumFn<FID> f{bases, vals, start, end, counters, extra, buckets, partials, sh};
    launch(f, slots);
  }
  // the block-level small-MSM path (curve_quad.hpp k_small_accum) is device-only: the emulation keeps the task path
  uint32_t small_chunk() const { return 0; }
  template <int FID>
  void launch_small_accum(const AffineW*, const uint32_t*, const uint32_t*, const uint32_t*, XYZZW*, XYZZW*, uint32_t, uint32_t, uint32_t) {}
  template <int FID>
  void launch_small_combine(const AffineW*, const uint32_t*, const uint32_t*, const uint32_t*, XYZZW*, XYZZW*, uint32_t, uint32_t) {}
  template <int FID>
  void launch_fold(const uint32_t* counters, const HeavyRec* heavy, XYZZW* partials, XYZZW* buckets, uint32_t T,
                   uint32_t cap, uint32_t groups) {
    FoldFn<FID> f{counters, heavy, partials, buckets, T, cap, groups};
    launch(f, groups * T);
  }
  template <int FID> const XYZZW* reduce_tree(const XYZZW* buckets, const MsmShape& sh, const uint32_t*, bool*) {  // one launch per level
    const XYZZW* D = buckets;
    const XYZZW* Y = buckets;
    uint32_t n_in = sh.M, first = 1;
    while (n_in > 1) {
      const uint32_t half = n_in / 2, pairs = sh.WB * half, padded = (pairs + 63u) & ~63u;
      XYZZW* Do = alloc<XYZZW>(pairs);
      XYZZW* Yo = alloc<XYZZW>(pairs);
      ReducePairFn<FID> f{D, Y, Do, Yo, n_in, pairs, padded, first};
      launch(f, 2 * padded);
      D = Do, Y = Yo, n_in = half, first = 0;
    }
    return Y;
  }
  template <int FID>
  void launch_big_all(const uint32_t* counters, const HeavyRec* big, const uint32_t* items, const uint32_t* gbase,
                      const XYZZL* bucket_raw, XYZZL* partial_raw, XYZZW* buckets, uint32_t*, uint32_t*, uint32_t slice) {
    // the item numbering of the plan step is checked here (the device kernel depends on it): every bucket's slices, once each
    std::vector<uint32_t> seen(counters[4], 0);
    for (uint32_t i = 0; i < counters[6]; i++) {
      const uint32_t h = items[i];
      if (h >= counters[4] || i < big[h].pad || i - big[h].pad >= (big[h].cnt + slice - 1) / slice) { fprintf(stderr, "big-bucket item %u is wrong\n", i); abort(); }
      seen[h]++;
    }
    for (uint32_t h = 0; h < counters[4]; h++)
      if (seen[h] != (big[h].cnt + slice - 1) / slice) { fprintf(stderr, "big bucket %u: %u items\n", h, seen[h]); abort(); }
    // group numbering: disjoint ranges of ceil(slices / 32) counters per bucket, counters[7] in total
    std::vector<uint32_t> gseen(counters[7], 0);
    for (uint32_t h = 0; h < counters[4]; h++) {
      const uint32_t ng = ((big[h].cnt + slice - 1) / slice + SegPlan::kBigGroup - 1) / SegPlan::kBigGroup;
      for (uint32_t g = 0; g < ng; g++) {
        if (gbase[h] + g >= counters[7] || gseen[gbase[h] + g]++) { fprintf(stderr, "big bucket %u: group %u is wrong\n", h, g); abort(); }
      }
    }
    BigBucketFn<FID> f{counters, big, bucket_raw, partial_raw, buckets};
    launch(f, counters[4] + 1);
  }
  // segment-balanced accumulate (msm_seg.hpp): an odd lane count so segments straddle bucket boundaries everywhere
  template <int FID> uint32_t seg_lanes(size_t) { return g_seg_lanes; }
  template <int FID>
  void launch_fold_raw(const uint32_t* counters, const HeavyRec* list, XYZZL* partial_raw, uint32_t T, uint32_t cap,
                       uint32_t groups, uint32_t use_big) {
    FoldRawFn<FID> f{counters, list, partial_raw, T, cap, groups, use_big};
    launch(f, groups * T);
  }
  template <int FID>
  void launch_final_seg(const uint32_t* start, const uint32_t* end, const uint32_t* total_p, const XYZZL* bucket_raw,
                        const XYZZL* partial_raw, XYZZW* buckets, uint32_t nbuckets, uint32_t lanes, uint32_t min_seg,
                        uint32_t heavy_above) {
    FinalSegFn<FID> f{start, end, total_p, bucket_raw, partial_raw, buckets, nbuckets, lanes, min_seg, heavy_above};
    launch(f, nbuckets);
  }
  void sort_pairs(uint32_t* k_in, uint32_t* k_out, uint32_t* v_in, uint32_t* v_out, size_t total, uint32_t bits) {
    std::vector<uint32_t> idx(total);
    std::iota(idx.begin(), idx.end(), 0u);
    uint32_t mask = bits >= 32 ? 0xffffffffu : ((1u << bits) - 1u);
    std::stable_sort(idx.begin(), idx.end(),
                     [&](uint32_t a, uint32_t b) { return (k_in[a] & mask) < (k_in[b] & mask); });
    for (size_t i = 0; i < total; i++) {
      k_out[i] = k_in[idx[i]];
      v_out[i] = v_in[idx[i]];
    }
  }
  void d2h(void* dst, const void* src, size_t bytes) { memcpy(dst, src, bytes); }
  void d2h_split(void* d1, size_t b1, void* d2, size_t b2, const void* src) {
    memcpy(d1, src, b1);
    memcpy(d2, (const char*)src + b1, b2);
  }
  void sync() {}
  void mark(const char*) {}
};


template <int CID>
static int emul_msm_t(const uint8_t* scalars, const uint8_t* bases_xy64, size_t n, uint32_t u64_bits,
                      uint32_t u64_mode, uint32_t scalars_mont, uint32_t force_c, uint8_t* out, uint8_t* inf,
                      uint32_t pre_c = 0, size_t n_key = 0, size_t pre_offset = 0) {
  using C = CurveT<CID>;
  constexpr int BF = C::BF, SF = C::SF;
  XYZZ<BF> r = XYZZ<BF>::identity();
  if (n != 0 && !(u64_mode && u64_bits == 0)) {
    // precomputed-table mode: bases_xy64 holds the whole key (n_key points); the MSM uses [pre_offset, +n)
    const size_t nb = pre_c ? n_key : n;
    const uint32_t Wt = pre_c ? (FpParams<SF>::BITS + 1 + pre_c - 1) / pre_c : 1;
    std::vector<AffineW> b(nb * Wt);
    for (size_t i = 0; i < nb; i++) {
      Affine<BF> a;
      a.x = fp_from_bytes<BF>(bases_xy64 + 64 * i).to_internal().canon();
      a.y = fp_from_bytes<BF>(bases_xy64 + 64 * i + 32).to_internal().canon();
      a.store(b[i]);
    }
    MsmArgs a;
    if (pre_c) {
      HostEmulBackend pb;
      PrecompFn<BF> pf{b.data(), (uint32_t)nb, pre_c, Wt};
      pb.launch(pf, (uint32_t)nb);
      a.pre_stride = (uint32_t)nb;
      a.pre_offset = (uint32_t)pre_offset;
      a.pre_c = pre_c;
    }
    a.scalars = (const uint32_t*)scalars;
    a.bases = b.data();
    a.n = (uint32_t)n;
    a.scalars_mont = scalars_mont;
    a.u64_bits = u64_mode ? u64_bits : 0;
    a.force_c = force_c;
    a.seg_min_total = g_seg_min_total;
    a.seg_min_len = 3;
    HostEmulBackend be;
    XYZZW wsum[260];
    uint32_t err = 0;
    MsmShape sh = msm_pipeline<HostEmulBackend, BF, SF>(be, a, FpParams<SF>::BITS, wsum, &err);
    if (err) return -(int)err - 100;
    r = combine_windows<BF>(wsum, sh);
  }
  xyzz_to_xy64<BF>(r, out, inf);
  return 0;
}

// fused batch over the tables of one key (MsmArgs::batch_*): k vectors, vector j over key[offset .. offset + lens[j])
template <int CID>
static int emul_msm_batch_t(const uint8_t* const* vecs, const size_t* lens, size_t k, const uint8_t* key_xy64, size_t n_key,
                            size_t offset, uint32_t pre_c, uint32_t scalars_mont, uint8_t* out, uint8_t* inf) {
  using C = CurveT<CID>;
  constexpr int BF = C::BF, SF = C::SF;
  const uint32_t Wt = (FpParams<SF>::BITS + 1 + pre_c - 1) / pre_c;
  std::vector<AffineW> b(n_key * Wt);
  for (size_t i = 0; i < n_key; i++) {
    Affine<BF> a;
    a.x = fp_from_bytes<BF>(key_xy64 + 64 * i);
    a.y = fp_from_bytes<BF>(key_xy64 + 64 * i + 32);
    if (!a.is_identity()) {
      a.x = a.x.to_internal().canon();
      a.y = a.y.to_internal().canon();
    }
    a.store(b[i]);
  }
  {
    HostEmulBackend pb;
    PrecompFn<BF> pf{b.data(), (uint32_t)n_key, pre_c, Wt};
    pb.launch(pf, (uint32_t)n_key);
  }
  std::vector<uint32_t> off(k + 1, 0);
  std::vector<const uint32_t*> ptrs(k);
  for (size_t j = 0; j < k; j++) {
    off[j + 1] = off[j] + (uint32_t)lens[j];
    ptrs[j] = (const uint32_t*)vecs[j];
  }
  XYZZW wsum[260];
  for (size_t j = 0; j < 260; j++) XYZZ<BF>::identity().store(wsum[j]);
  if (off[k]) {
    MsmArgs a;
    a.scalars = nullptr;
    a.bases = b.data();
    a.n = off[k];
    a.scalars_mont = scalars_mont;
    a.u64_bits = 0;
    a.force_c = 0;
    a.pre_stride = (uint32_t)n_key;
    a.pre_offset = (uint32_t)offset;
    a.pre_c = pre_c;
    a.seg_min_total = g_seg_min_total;
    a.seg_min_len = 3;
    a.batch_k = (uint32_t)k;
    a.batch_off = off.data();
    a.batch_vec = ptrs.data();
    HostEmulBackend be;
    uint32_t err = 0;
    MsmShape sh = msm_pipeline<HostEmulBackend, BF, SF>(be, a, FpParams<SF>::BITS, wsum, &err);
    if (err) return -(int)err - 100;
    if (!partition_supported(sh, true)) return -3;  // the test means to exercise the partition's batch keys
  }
  for (size_t j = 0; j < k; j++) xyzz_to_xy64<BF>(XYZZ<BF>::load(wsum[j]), out + 64 * j, inf + j);
  return 0;
}

// inputs: 256-bit integers (any value the test wants, e.g. up to 8p); outputs canonicalised
template <int FID> static void fp_op_t(int op, const uint8_t* a, const uint8_t* b, uint8_t* out) {
  using F = Fp<FID>;
  F x = fp_from_bytes<FID>(a), y = fp_from_bytes<FID>(b), r;
  switch (op) {
    case 0: r = (x * y).canon(); break;                 // x*y*2^-261
    case 1: r = (x + y).norm().canon(); break;          // lazy add
    case 2: r = F::sub2(x, y).norm().canon(); break;    // x - y + 2p   (y < 2p)
    case 3: r = x.to_internal().canon(); break;         // x*2^261
    case 4: r = x.to_canonical(); break;                // x*2^-261
    case 5: r = x.inv().canon(); break;                 // internal-form inverse
    case 6: r = F::sub8(x, y).norm().canon(); break;    // x - y + 8p   (y < 8p)
    case 7: r = x.sqr().canon(); break;
    case 8: r = x.mont256_to_internal().canon(); break; // x*2^5 (halo2 Montgomery -> internal)
    case 9: r = x.mont256_to_canonical(); break;        // x*2^-256
    case 10: r = F::sub4(x, y).norm().canon(); break;
    case 11: r = x.canon(); break;
    case 12: { r = F::zero(); r.l[0] = x.norm().maybe_zero_mod_p() ? 1u : 0u; } break;
    // latency-oriented products (separated operand scanning): must equal the chained ones limb for limb
    case 13: { F u = F::template mulx<true>(x, y), v = x * y; for (int i = 0; i < 9; i++) if (u.l[i] != v.l[i]) u = F::zero(); r = u.canon(); } break;
    case 14: { F u = F::template sqrx<true>(x), v = x.sqr(); for (int i = 0; i < 9; i++) if (u.l[i] != v.l[i]) u = F::zero(); r = u.canon(); } break;
    case 15: { F u = F::template mul_addx<true>(x, y, y, x), v = F::mul_add(x, y, y, x); for (int i = 0; i < 9; i++) if (u.l[i] != v.l[i]) u = F::zero(); r = u.canon(); } break;
    // Fp::dot (round 3): up to four products under one reduction, against the same products one by one
    case 16: { const F a4[4] = {x, y, x, y}, b4[4] = {y, x, x, y}; r = F::template dot<4>(a4, b4).canon(); } break;   // 2xy + x^2 + y^2
    case 17: { const F a3[3] = {x, x, y}, b3[3] = {y, x, y}; r = F::template dot<3>(a3, b3).canon(); } break;          // xy + x^2 + y^2
    case 18: { const F a1[1] = {x}, b1[1] = {y}; F u = F::template dot<1>(a1, b1), v = x * y; for (int i = 0; i < 9; i++) if (u.l[i] != v.l[i]) u = F::zero(); r = u.canon(); } break;
    default: r = F::zero();
  }
  fp_to_bytes(r, out);
}

// The partition kernels alone (no curve arithmetic) against DigitsFn + std::sort: same multiset of (row | sign) words
// per bucket, consistent start / end, total = number of non-zero digits.  grid_override forces the grid-stride loops.
static int partition_check(const uint8_t* scalars, size_t n, uint32_t c, uint32_t u64_bits, uint32_t grid_override,
                           uint32_t stride, uint32_t offset, uint32_t ct_width) {
  constexpr int SF = F_BN254_FR;
  const uint32_t bits = u64_bits ? u64_bits : (uint32_t)FpParams<SF>::BITS;
  const bool table_mode = stride != 0;  // stride 0: plain keys, one bucket set per window (key = w * M + |d| - 1)
  MsmShape sh = table_mode ? make_shape((uint32_t)n, bits, 0, c) : make_shape((uint32_t)n, bits, c, 0);
  if (!partition_supported(sh, table_mode)) return -2;
  const size_t total = sh.total;
  PartArgs<SF> pa;
  PartBufs& pb = pa.b;
  pb.ps = make_part_shape(sh, table_mode);
  if (grid_override) pb.ps.grid1 = grid_override;
  pb.nbuckets = sh.nbuckets;
  std::vector<uint32_t> start(sh.nbuckets + 1), end(sh.nbuckets + 1), ctr(2048 + 3 * kTabStride + 1 + 2 * (size_t)sh.nbuckets),
      vals(total + 1), ent_val(pb.ps.ent_cap), keys(total + 1), kvals(total + 1);
  std::vector<uint16_t> ent_lo(pb.ps.ent_cap);
  uint32_t err = 0, tot = 0;
  DigitSrc<SF> src{(const uint32_t*)scalars, nullptr, &err, sh, 0, u64_bits, stride, offset, nullptr, 0};
  pa.src = src;
  pb.hist_hi = ctr.data();
  pb.cur_hi = ctr.data() + 1024;
  pb.tab = ctr.data() + 2048;
  pb.bucket_cnt = ctr.data() + 2048 + 3 * kTabStride + 1;
  pb.bucket_cur = pb.bucket_cnt + sh.nbuckets;
  pb.ent_val = ent_val.data();
  pb.ent_lo = ent_lo.data();
  pb.start = start.data();
  pb.end = end.data();
  pb.vals = vals.data();
  pb.total_out = &tot;
  HostEmulBackend be;
  pb.single_bin = pb.ps.nhi == 1 ? 1u : 0u;
  pa.b.single_bin = pb.single_bin;
  auto l1 = [&](auto hist, auto part) {
    if (pb.single_bin) {  // as msm_pipeline.hpp: placing pass first, the tile table from its cursor, no counting pass
      be.launch_kernel(part, pb.ps.grid1, pb.ps.bs1, pa);
      PartBufs pt = pb;
      pt.hist_hi = pb.cur_hi;
      if (pb.ps.big) be.launch_kernel(&k_tiles<true>, 1u, 1024u, pt);
      else be.launch_kernel(&k_tiles<false>, 1u, 1024u, pt);
      return;
    }
    be.launch_kernel(hist, pb.ps.grid1, pb.ps.bs1, pa);
    if (pb.ps.big) be.launch_kernel(&k_tiles<true>, 1u, 1024u, pb);
    else be.launch_kernel(&k_tiles<false>, 1u, 1024u, pb);
    be.launch_kernel(part, pb.ps.grid1, pb.ps.bs1, pa);
  };
  const uint32_t cc = ct_width ? c : 0;  // compile-time-width instantiation (when there is one) or the run-time one
  if (pb.ps.big) {
    if (cc == 20) l1(&k_hist_hi<SF, 20, true>, &k_part_hi<SF, 20, true>);
    else if (cc == 16) l1(&k_hist_hi<SF, 16, true>, &k_part_hi<SF, 16, true>);
    else if (cc == 15) l1(&k_hist_hi<SF, 15, true>, &k_part_hi<SF, 15, true>);
    else l1(&k_hist_hi<SF, 0, true>, &k_part_hi<SF, 0, true>);
    be.launch_kernel(&k_hist_lo<true>, pb.ps.tiles_cap, kTileThreads, pb);
    be.launch_kernel(&k_part_lo<true>, pb.ps.tiles_cap, kTileThreads, pb);
  } else {
    if (cc == 17) l1(&k_hist_hi<SF, 17, false>, &k_part_hi<SF, 17, false>);
    else if (cc == 16) l1(&k_hist_hi<SF, 16, false>, &k_part_hi<SF, 16, false>);
    else if (cc == 15) l1(&k_hist_hi<SF, 15, false>, &k_part_hi<SF, 15, false>);
    else if (cc == 8) l1(&k_hist_hi<SF, 8, false>, &k_part_hi<SF, 8, false>);
    else l1(&k_hist_hi<SF, 0, false>, &k_part_hi<SF, 0, false>);
    be.launch_kernel(&k_hist_lo<false>, pb.ps.tiles_cap, kTileThreads, pb);
    be.launch_kernel(&k_part_lo<false>, pb.ps.tiles_cap, kTileThreads, pb);
  }
  // reference: materialised (key, val) pairs grouped by key
  uint32_t err2 = 0;
  DigitSrc<SF> src2 = src;
  src2.err = &err2;
  DigitsFn<SF> df{src2, keys.data(), kvals.data()};
  for (uint32_t i = 0; i < n; i++) df(i);
  if (err != err2) return -3;
  std::vector<std::vector<uint32_t>> ref(sh.nbuckets);
  size_t nz = 0;
  for (size_t e = 0; e < total; e++)
    if (keys[e] < sh.nbuckets) {
      ref[keys[e]].push_back(kvals[e]);
      nz++;
    }
  if (tot != nz) return -4;
  uint32_t run = 0;
  for (uint32_t k = 0; k < sh.nbuckets; k++) {
    if (start[k] != run || end[k] != run + ref[k].size()) return -5;
    std::vector<uint32_t> got(vals.begin() + start[k], vals.begin() + end[k]);
    std::sort(got.begin(), got.end());
    std::sort(ref[k].begin(), ref[k].end());
    if (got != ref[k]) return -6;
    run = end[k];
  }
  if (start[sh.nbuckets] != run || end[sh.nbuckets] != run) return -7;
  return 0;
}

// window tables built in one pass (PrecompFn) and in two passes with a shared inversion (PrecompDblFn + PrecompNormFn):
// byte-identical?
template <int CID> static int precomp_check_t(const uint8_t* key_xy64, size_t n, uint32_t c) {
  using C = CurveT<CID>;
  constexpr int BF = C::BF;
  const uint32_t W = (FpParams<C::SF>::BITS + 1 + c - 1) / c;
  std::vector<AffineW> a(n * W), b(n * W);
  for (size_t i = 0; i < n; i++) {
    Affine<BF> p;
    p.x = fp_from_bytes<BF>(key_xy64 + 64 * i);
    p.y = fp_from_bytes<BF>(key_xy64 + 64 * i + 32);
    if (!p.is_identity()) {
      p.x = p.x.to_internal().canon();
      p.y = p.y.to_internal().canon();
    }
    p.store(a[i]);
    b[i] = a[i];
  }
  HostEmulBackend be;
  PrecompFn<BF> f0{a.data(), (uint32_t)n, c, W};
  be.launch(f0, (uint32_t)n);
  const size_t chunk = n / 2 + 1;  // two chunks: exercises i0 != 0
  std::vector<XYZZL> raw((W - 1) * chunk);
  std::vector<uint32_t> pref((W - 1) * chunk * 8);
  for (size_t i0 = 0; i0 < n; i0 += chunk) {
    const uint32_t m = (uint32_t)(n - i0 < chunk ? n - i0 : chunk);
    PrecompDblFn<BF> f1{b.data(), raw.data(), (uint32_t)i0, m, c, W};
    be.launch(f1, m);
    PrecompNormFn<BF> f2{raw.data(), pref.data(), b.data(), (uint32_t)n, (uint32_t)i0, m, W};
    be.launch(f2, m);
  }
  return memcmp(a.data(), b.data(), sizeof(AffineW) * n * W) == 0 ? 0 : -1;
}

extern "C" {

int emul_precomp_check(int curve, const uint8_t* key_xy64, size_t n, uint32_t c) {
  switch (curve) {
    case 0: return precomp_check_t<0>(key_xy64, n, c);
    case 1: return precomp_check_t<1>(key_xy64, n, c);
    case 2: return precomp_check_t<2>(key_xy64, n, c);
    case 3: return precomp_check_t<3>(key_xy64, n, c);
  }
  return -2;
}

unsigned long long emul_block_kernel_launches() { return g_block_kernel_launches; }  // the partition's kernels (tests: which path ran)
void emul_set_seg_min_total(uint32_t v) { g_seg_min_total = v; }
void emul_set_seg_lanes(uint32_t v) { g_seg_lanes = v; }

int emul_partition_check(const uint8_t* scalars, size_t n, uint32_t c, uint32_t u64_bits, uint32_t grid_override,
                         uint32_t stride, uint32_t offset, uint32_t ct_width) {
  return partition_check(scalars, n, c, u64_bits, grid_override, stride, offset, ct_width);
}

int emul_msm(int curve, const uint8_t* scalars, const uint8_t* bases_xy64, size_t n, uint32_t u64_bits,
             uint32_t u64_mode, uint32_t scalars_mont, uint32_t force_c, uint8_t* out, uint8_t* inf) {
  switch (curve) {
    case 0: return emul_msm_t<0>(scalars, bases_xy64, n, u64_bits, u64_mode, scalars_mont, force_c, out, inf);
    case 1: return emul_msm_t<1>(scalars, bases_xy64, n, u64_bits, u64_mode, scalars_mont, force_c, out, inf);
    case 2: return emul_msm_t<2>(scalars, bases_xy64, n, u64_bits, u64_mode, scalars_mont, force_c, out, inf);
    case 3: return emul_msm_t<3>(scalars, bases_xy64, n, u64_bits, u64_mode, scalars_mont, force_c, out, inf);
  }
  return -1;
}

// precomputed-table mode over a key of n_key points, MSM over key[offset .. offset+n)
int emul_msm_precomp(int curve, const uint8_t* scalars, const uint8_t* key_xy64, size_t n_key, size_t offset, size_t n,
                     uint32_t u64_bits, uint32_t u64_mode, uint32_t pre_c, uint8_t* out, uint8_t* inf) {
  switch (curve) {
    case 0: return emul_msm_t<0>(scalars, key_xy64, n, u64_bits, u64_mode, 0, 0, out, inf, pre_c, n_key, offset);
    case 1: return emul_msm_t<1>(scalars, key_xy64, n, u64_bits, u64_mode, 0, 0, out, inf, pre_c, n_key, offset);
    case 2: return emul_msm_t<2>(scalars, key_xy64, n, u64_bits, u64_mode, 0, 0, out, inf, pre_c, n_key, offset);
    case 3: return emul_msm_t<3>(scalars, key_xy64, n, u64_bits, u64_mode, 0, 0, out, inf, pre_c, n_key, offset);
  }
  return -1;
}

int emul_msm_batch(int curve, const uint8_t* const* vecs, const size_t* lens, size_t k, const uint8_t* key_xy64,
                   size_t n_key, size_t offset, uint32_t pre_c, uint32_t scalars_mont, uint8_t* out, uint8_t* inf) {
  switch (curve) {
    case 0: return emul_msm_batch_t<0>(vecs, lens, k, key_xy64, n_key, offset, pre_c, scalars_mont, out, inf);
    case 1: return emul_msm_batch_t<1>(vecs, lens, k, key_xy64, n_key, offset, pre_c, scalars_mont, out, inf);
    case 2: return emul_msm_batch_t<2>(vecs, lens, k, key_xy64, n_key, offset, pre_c, scalars_mont, out, inf);
    case 3: return emul_msm_batch_t<3>(vecs, lens, k, key_xy64, n_key, offset, pre_c, scalars_mont, out, inf);
  }
  return -1;
}

int emul_fp_op(int fid, int op, const uint8_t* a, const uint8_t* b, uint8_t* out) {
  switch (fid) {
    case 0: fp_op_t<0>(op, a, b, out); return 0;
    case 1: fp_op_t<1>(op, a, b, out); return 0;
    case 2: fp_op_t<2>(op, a, b, out); return 0;
    case 3: fp_op_t<3>(op, a, b, out); return 0;
  }
  return -1;
}

// generator of curve `cid` as canonical x||y (checks the constants in curves.hpp)
int emul_generator(int curve, uint8_t* out) {
  switch (curve) {
    case 0: memcpy(out, CurveT<0>::GX, 32); memcpy(out + 32, CurveT<0>::GY, 32); return 0;
    case 1: memcpy(out, CurveT<1>::GX, 32); memcpy(out + 32, CurveT<1>::GY, 32); return 0;
    case 2: memcpy(out, CurveT<2>::GX, 32); memcpy(out + 32, CurveT<2>::GY, 32); return 0;
    case 3: memcpy(out, CurveT<3>::GX, 32); memcpy(out + 32, CurveT<3>::GY, 32); return 0;
  }
  return -1;
}
}
