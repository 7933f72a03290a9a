// curve_impl.hpp -- everything that is instantiated per curve; included by the four curve_*.hip TUs only.
//
// Replaces, behind the reference's DlogGroupExt / CommitmentEngineTrait seam:
//   /root/reference/src/provider/msm.rs:225-419,469-503  (msm, msm_small, msm_small_with_max_num_bits)
//   halo2curves::msm::msm_best (called at msm.rs:411,500)
//   /root/reference/src/provider/pedersen.rs:263-270, hyperkzg.rs:584-591 (commit = msm + h*r)
#pragma once
#include <array>

#include "runtime.hpp"

namespace nmx {

// ---------------------------------------------------------------------------------------------------
// small device helpers
// ---------------------------------------------------------------------------------------------------
template <int FID> struct ToInternalFn {  // ABI form -> internal form (canonical residue), in place, one element per lane
  uint32_t* v;         // 8 words per element
  uint32_t from_mont;  // 1: input is halo2curves Montgomery (x * 2^256); 0: canonical integer
  NMX_HD void operator()(uint32_t i) const {
    uint32_t* w = v + 8 * (size_t)i;
    Fp<FID> f = Fp<FID>::from_words(w);
    f = from_mont ? f.mont256_to_internal() : f.to_internal();
    f.canon().to_words(w);  // 0 -> 0: the identity encoding (0, 0) is preserved
  }
};
// NMX_BASES_VALIDATE: raw ABI-form point i -> error bit if a coordinate is >= p or y^2 != x^3 + b (read_points,
// /root/reference/src/provider/ptau.rs:372-391: read_raw rejects non-canonical coordinates, then is_on_curve; the
// identity (0, 0) passes is_on_curve).  Runs BEFORE the in-place conversion.
template <int CID> struct ValidateFn {
  using C = CurveT<CID>;
  const uint32_t* v;  // n x 16 words
  uint32_t from_mont;
  uint32_t* err;
  NMX_HD void operator()(uint32_t i) const {
    using F = Fp<C::BF>;
    const uint32_t* w = v + 16 * (size_t)i;
    uint32_t any = 0;
    for (int j = 0; j < 16; j++) any |= w[j];
    if (!any) return;
    bool ok = F::words_lt_p(w) && F::words_lt_p(w + 8);
    if (ok) {
      F x = F::from_words(w), y = F::from_words(w + 8);
      x = (from_mont ? x.mont256_to_internal() : x.to_internal());
      y = (from_mont ? y.mont256_to_internal() : y.to_internal());
      uint32_t bw[8];
      for (int j = 0; j < 8; j++) bw[j] = C::B[j];
      F b = F::from_words(bw).to_internal();
      F lhs = y.sqr();
      F rhs = (x.sqr() * x + b).norm();
      ok = F::eq_mod_p(lhs, rhs);
    }
    if (!ok) nmx_atomic_or(err, 1u);
  }
};
struct AnyIdentityFn {  // does the key hold the identity encoding (all-zero x || y) anywhere?
  const uint32_t* v;  // n x 16 words
  uint32_t* flag;
  NMX_HD void operator()(uint32_t i) const {
    const uint32_t* w = v + 16 * (size_t)i;
    uint32_t any = 0;
#pragma unroll
    for (int j = 0; j < 16; j++) any |= w[j];
    if (!any) nmx_atomic_or(flag, 1u);
  }
};
static bool scan_identity(Ctx& c, const void* d, size_t n) {
  if (n == 0) return false;
  arena_reserve(c, 256);
  uint32_t* dflag = (uint32_t*)c.arena;
  HIPCHK(hipMemsetAsync(dflag, 0, 4, c.stream));
  DeviceBackend be(c, false, false);
  AnyIdentityFn f{(const uint32_t*)d, dflag};
  be.launch(f, (uint32_t)n);
  uint32_t h = 0;
  HIPCHK(hipMemcpyAsync(&h, dflag, 4, hipMemcpyDeviceToHost, c.stream));
  HIPCHK(hipStreamSynchronize(c.stream));
  return h != 0;
}
template <int CID> struct GenFn {  // P_i = (k0 + i) * G
  using C = CurveT<CID>;
  AffineW* out;
  uint64_t k0;
  NMX_HD void operator()(uint32_t i) const {
    uint32_t wx[8], wy[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
      wx[j] = C::GX[j];
      wy[j] = C::GY[j];
    }
    Fp<C::BF> gx = Fp<C::BF>::from_words(wx).to_internal().canon();
    Fp<C::BF> gy = Fp<C::BF>::from_words(wy).to_internal().canon();
    uint64_t k = k0 + i;
    XYZZ<C::BF> acc = XYZZ<C::BF>::identity();
    for (int b = 63; b >= 0; b--) {
      acc.dbl_in_place();
      if ((k >> b) & 1u) acc.add_affine(gx, gy);
    }
    acc.to_affine().store(out[i]);
  }
};

// ---------------------------------------------------------------------------------------------------
// one MSM on the device
// ---------------------------------------------------------------------------------------------------
static inline uint64_t shape_hash(const MsmArgs& a, const MsmCall& mc, size_t sbytes, uint32_t cid, uint32_t sbits,
                                  uint32_t seg_lanes) {
  const uint64_t v[11] = {((uint64_t)cid << 56) | ((uint64_t)sbits << 40) | seg_lanes, a.n, a.u64_bits | ((uint64_t)G.no_tree_fuse.load(std::memory_order_relaxed) << 32) | ((uint64_t)G.tree_threads.load(std::memory_order_relaxed) << 40) | ((uint64_t)a.big_slice << 44), a.force_c, a.force_lmax, a.force_fold_t, ((uint64_t)a.pre_stride << 8) | a.pre_c,
                          (uint64_t)(mc.gather_host != nullptr) | (mc.all_ones ? 2u : 0u) | (mc.scalars_device ? 4u : 0u) |
                              (a.no_partition ? 8u : 0u),
                          sbytes, a.seg_min_total, G.seg_lanes_override ^ ((uint64_t)a.seg_heavy_above << 32)};
  uint64_t h = 0x9e3779b97f4a7c15ull;
  for (uint64_t x : v) {
    h = (h ^ x) * 0xff51afd7ed558ccdull;
    h ^= h >> 31;
  }
  return h | 1u;  // never 0 (= "nothing remembered")
}

template <int CID>
static XYZZ<CurveT<CID>::BF> run_msm(Ctx& c, const void* d_bases, size_t n, const MsmCall& mc) {
  using C = CurveT<CID>;
  constexpr int BF = C::BF, SF = C::SF;
  XYZZ<BF> ident = XYZZ<BF>::identity();
  if (n == 0) return ident;                       // msm.rs:228
  if (mc.u64_mode && mc.u64_bits == 0) return ident;  // msm.rs:489
  const uint32_t sbits = FpParams<SF>::BITS;
  const size_t sbytes = mc.u64_mode ? 8 : 32;

  MsmArgs a;
  a.bases = d_bases;
  a.n = (uint32_t)n;
  a.scalars_mont = mc.scalars_mont ? 1u : 0u;
  a.u64_bits = mc.u64_mode ? mc.u64_bits : 0u;
  a.force_c = G.force_c.load(std::memory_order_relaxed);
  a.force_lmax = G.force_lmax;
  a.force_fold_t = G.force_fold_t;
  a.pre_stride = mc.pre_stride;
  a.pre_offset = mc.pre_offset;
  a.pre_c = mc.pre_c;
  a.bases_clean = mc.bases_clean ? 1u : 0u;
  a.no_partition = G.no_partition;
  a.seg_min_total = G.seg_min_total;
  a.seg_min_len = G.seg_min_len;
  a.seg_heavy_above = G.seg_heavy_above;
  a.big_slice = G.big_slice.load(std::memory_order_relaxed);
  a.hist_grid = G.hist_grid;
  a.hist_bs = G.hist_bs;
  // gathers in flight per lane: one is enough while the key's tables (W x 64 B per point) mostly hit the 256 MB Infinity Cache
  // and L2; from ~6 GiB of tables on the gather latency shows and a second row in flight pays (2^24: accumulate 17.6 ->
  // 16.0 ms; neutral at 2^22, slightly worse at 2^20 / 2^21: profiles/r02_msm_2p20/prefetch_depth.txt)
  const uint32_t pf_opt = G.accum_prefetch.load(std::memory_order_relaxed);
  a.accum_prefetch = pf_opt ? pf_opt
                                      : ((uint64_t)mc.pre_stride * 64u * ((FpParams<SF>::BITS + 1 + (mc.pre_c ? mc.pre_c : 16) - 1) /
                                                                         (mc.pre_c ? mc.pre_c : 16)) >= (6ull << 30) ? 2u : 1u);
  {
    uint32_t bits = a.u64_bits ? a.u64_bits : sbits;
    MsmShape sh = make_shape(a.n, bits, a.force_c, a.pre_stride ? a.pre_c : 0);
    // (the partition's intermediate arrays carry up to 2^20 entries of alignment slack on top of n * windows)
    require((uint64_t)n * sh.W < 0xfff00000ull && n < 0x7fffffffull, NMX_E_TOO_LARGE,
            "n * windows must be < 2^32");
  }
  c.wsum.resize(260);
  XYZZW* wsum = c.wsum.data();
  uint32_t err = 0;
  MsmShape sh{};
  const bool prof = G.profiling.load(std::memory_order_relaxed);
  // The dry pass only sizes the workspace; its answer is a function of the call's shape, remembered per context.
  const uint64_t shape_key = shape_hash(a, mc, sbytes, (uint32_t)CID, sbits, DeviceBackend(c, true, false).template seg_lanes<BF>((size_t)a.n * 16));
  // A remembered workspace size that turns out too small (a stale entry) is not fatal: forget it and size again.
  for (int attempt = 0; attempt < 2; attempt++) try {
  for (int pass = (c.shape_key == shape_key && c.shape_bytes <= c.cap) ? 1 : 0; pass < 2; pass++) {
    DeviceBackend be(c, pass == 0, prof);
    if (mc.gather_host) {
      uint32_t* d_g = be.alloc<uint32_t>(n);
      a.gather = d_g;
      if (pass == 1) HIPCHK(hipMemcpyAsync(d_g, mc.gather_host, n * 4, hipMemcpyHostToDevice, c.stream));
    }
    a.all_ones = mc.all_ones ? 1u : 0u;
    if (mc.all_ones) {
      a.scalars = nullptr;
    } else if (mc.scalars_device) {
      a.scalars = (const uint32_t*)mc.scalars;
    } else {
      uint32_t* d_s = be.alloc<uint32_t>(n * sbytes / 4);
      a.scalars = d_s;
      if (pass == 1)
        HIPCHK(hipMemcpyAsync(d_s, mc.scalars, n * sbytes, hipMemcpyHostToDevice, c.stream));
    }
    sh = msm_pipeline<DeviceBackend, BF, SF>(be, a, sbits, wsum, &err);
    if (pass == 0) {
      arena_reserve(c, be.used);
      c.shape_key = shape_key;
      c.shape_bytes = be.used;
    } else if (prof) {
      float st[kMaxMarks];
      int ns = 0;
      for (int i = 0; i + 1 < be.nmarks; i++) {
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, c.ev[i], c.ev[i + 1]));
        st[ns++] = ms;
      }
      prof_store(st, ns);
    }
  }
  break;
  } catch (const Fail& f) {
    if (attempt == 1 || f.msg != "workspace arena overflow") throw;
    (void)hipStreamSynchronize(c.stream);
    c.shape_key = 0;
  }
  require(!(err & ERR_SCALAR_RANGE), NMX_E_SCALAR_RANGE, "scalar >= field modulus");
  require(!(err & ERR_SMALL_RANGE), NMX_E_SMALL_RANGE, "small scalar >= 2^max_num_bits");
  auto t0 = std::chrono::steady_clock::now();
  XYZZ<BF> r = combine_windows<BF>(wsum, sh);
  if (prof) {
    auto t1 = std::chrono::steady_clock::now();
    prof_add_tail(std::chrono::duration<float, std::milli>(t1 - t0).count());
  }
  return r;
}

template <int CID> static void write_result(const XYZZ<CurveT<CID>::BF>& r, uint32_t flags, uint8_t* out,
                                            uint8_t* is_inf) {
  if (flags & NMX_OUT_PARTIAL) {
    XYZZW w;
    r.store(w);
    memcpy(out, w.w, 128);
    if (is_inf) *is_inf = r.is_identity() ? 1 : 0;
  } else {
    xyzz_to_xy64<CurveT<CID>::BF>(r, out, is_inf);
  }
}

// upload (or adopt) a base array, returning a Montgomery-form device copy owned by the caller
// number of tables / window width for a key of n points (0 tables = no precomputation)
template <int CID> static void table_shape(size_t n, uint32_t flags, uint32_t* pre_c, uint32_t* pre_W) {
  *pre_c = 0;
  *pre_W = 0;
  if (!(flags & NMX_BASES_PRECOMPUTE) || n < G.precomp_min_n) return;
  const uint32_t fc = G.force_c.load(std::memory_order_relaxed);
  uint32_t c = fc ? fc : choose_c_precomp((uint32_t)n, FpParams<CurveT<CID>::SF>::BITS);
  uint32_t W = (FpParams<CurveT<CID>::SF>::BITS + 1 + c - 1) / c;
  if ((uint64_t)W * n >= (1ull << 31)) return;  // table index must fit 31 bits
  *pre_c = c;
  *pre_W = W;
}
static inline void apply_table_limit(size_t n, uint32_t* pre_c, uint32_t* pre_W) {  // option max_table_mib
  const size_t limit = G.max_table_bytes.load(std::memory_order_relaxed);
  if (*pre_W && limit && n * 64 * (size_t)*pre_W > limit) {
    *pre_c = *pre_W = 0;
    note_table_fallback();
  }
}
struct ScratchFree {  // registration-time scratch (not the per-call arena: it can be gigabytes, and it is needed once)
  void* p = nullptr;
  ~ScratchFree() {
    if (p) (void)hipFree(p);
  }
};
template <int CID> static void build_tables(Ctx& c, void* d, size_t n, uint32_t pre_c, uint32_t pre_W) {
  if (!pre_W) return;
  constexpr int BF = CurveT<CID>::BF;
  DeviceBackend be(c, false, false);
  if (pre_W < 3 || n < 1024) {  // small keys: one pass, one inversion per table point
    PrecompFn<BF> f{(AffineW*)d, (uint32_t)n, pre_c, pre_W};
    be.launch(f, (uint32_t)n);
    return;
  }
  // two passes with a shared inversion per key point, 2^20 key points at a time (scratch: 176 B per table point)
  const size_t chunk = n < ((size_t)1 << 20) ? n : ((size_t)1 << 20);
  const size_t raw_bytes = (size_t)(pre_W - 1) * chunk * sizeof(XYZZL), pref_bytes = (size_t)(pre_W - 1) * chunk * 32;
  ScratchFree sc;
  HIPCHK(hipMalloc(&sc.p, raw_bytes + pref_bytes));
  for (size_t i0 = 0; i0 < n; i0 += chunk) {
    const uint32_t m = (uint32_t)(n - i0 < chunk ? n - i0 : chunk);
    PrecompDblFn<BF> f1{(const AffineW*)d, (XYZZL*)sc.p, (uint32_t)i0, m, pre_c, pre_W};
    be.launch(f1, m);
    PrecompNormFn<BF> f2{(const XYZZL*)sc.p, (uint32_t*)((char*)sc.p + raw_bytes), (AffineW*)d, (uint32_t)n, (uint32_t)i0, m, pre_W};
    be.launch(f2, m);
  }
  HIPCHK(hipStreamSynchronize(c.stream));  // the scratch is freed on return
}

template <int CID>
static void upload_bases(Ctx& c, BaseSet& bs, const void* src, uint32_t flags, const BaseFill* fill) {
  constexpr int BF = CurveT<CID>::BF;
  const size_t n = bs.n;
  uint32_t* pre_c = &bs.pre_c;
  uint32_t* pre_W = &bs.pre_W;
  void* d = nullptr;
  // lane counts below are 32-bit: 2 * n conversions, n validations (the file-backed entry points check this too)
  require(n < (1ull << 31), NMX_E_TOO_LARGE, "key too large (n must be < 2^31)");
  table_shape<CID>(n, flags, pre_c, pre_W);
  bs.any_identity = false;
  if (n == 0) return;
  // Degrade instead of failing when the window tables do not fit (the reference supports pruned .ptau
  // keys up to 2^28 points, README.md:130-138 -- 13 table copies of those are ~208 GiB): over the configured limit, or
  // hipMalloc out of memory -> the key alone, and MSMs over it take the plain path (W bucket sets, no tables).
  apply_table_limit(n, pre_c, pre_W);
  size_t alloc = n * 64 * (*pre_W ? *pre_W : 1);
  hipError_t me = hipMalloc(&d, alloc);
  if (me == hipErrorOutOfMemory && *pre_W) {
    (void)hipGetLastError();
    *pre_c = *pre_W = 0;
    note_table_fallback();
    alloc = n * 64;
    me = hipMalloc(&d, alloc);
  }
  if (me != hipSuccess) {
    (void)hipGetLastError();
    throw Fail{NMX_E_HIP, std::string("hipMalloc of the key: ") + hipGetErrorString(me)};
  }
  try {
    if (fill) (*fill)(d, c.stream);
    else
      HIPCHK(hipMemcpyAsync(d, src, n * 64,
                            (flags & NMX_BASES_DEVICE) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice,
                            c.stream));
    if ((flags & NMX_BASES_VALIDATE) && !(flags & NMX_BASES_INTERNAL)) {
      arena_reserve(c, 256);
      uint32_t* derr = (uint32_t*)c.arena;
      HIPCHK(hipMemsetAsync(derr, 0, 4, c.stream));
      DeviceBackend be(c, false, false);
      ValidateFn<CID> f{(const uint32_t*)d, (flags & NMX_BASES_MONT) ? 1u : 0u, derr};
      be.launch(f, (uint32_t)n);
      uint32_t herr = 0;
      HIPCHK(hipMemcpyAsync(&herr, derr, 4, hipMemcpyDeviceToHost, c.stream));
      HIPCHK(hipStreamSynchronize(c.stream));
      require(herr == 0, NMX_E_POINT, "PointNotOnCurve: a loaded point is not canonical or not on the curve");
    }
    if (!(flags & NMX_BASES_INTERNAL)) {
      DeviceBackend be(c, false, false);
      ToInternalFn<BF> f{(uint32_t*)d, (flags & NMX_BASES_MONT) ? 1u : 0u};
      be.launch(f, (uint32_t)(2 * n));
    }
    try {
      build_tables<CID>(c, d, n, *pre_c, *pre_W);
    } catch (const Fail& f) {  // the registration-time scratch (176 B per table point of a 2^20-point chunk) did not fit
      if (f.msg.find("out of memory") == std::string::npos) throw;
      (void)hipStreamSynchronize(c.stream);
      *pre_c = *pre_W = 0;  // the table area stays allocated but unused: MSMs take the plain path
      note_table_fallback();
    }
    bs.any_identity = scan_identity(c, d, n);  // also the stream sync that ends the upload
  } catch (...) {
    (void)hipFree(d);
    throw;
  }
  bs.d = d;
  bs.alloc_bytes = alloc;  // what the cache budget must count: the table area stays allocated when build_tables fell back
}

// MSM over bs[offset, offset + n): through the key's window tables when it has them and n is large enough
template <int CID>
static XYZZ<CurveT<CID>::BF> run_msm_key(Ctx& c, const BaseSet& bs, size_t offset, size_t n, MsmCall mc) {
  const uint32_t fc = G.force_c.load(std::memory_order_relaxed);
  if (bs.pre_W && n >= G.precomp_min_n && (fc == 0 || fc == bs.pre_c)) {
    mc.pre_stride = (uint32_t)bs.n;
    mc.pre_offset = (uint32_t)offset;
    mc.pre_c = bs.pre_c;
    mc.bases_clean = !bs.any_identity;
    return run_msm<CID>(c, bs.d, n, mc);
  }
  mc.bases_clean = !bs.any_identity;
  return run_msm<CID>(c, (const char*)bs.d + offset * 64, n, mc);
}
// Fused batch over the key's tables: every vector's digits go through one partition / accumulate / reduction run, vector j
// owning bucket set j.  The fixed latency of a run (~0.2-0.3 ms of dependent point additions) is paid once, not k times.
template <int CID> static uint32_t batch_limit_for(const BaseSet& bs) {
  if (!bs.pre_W || G.no_batch_fuse) return 0;
  const uint32_t fc = G.force_c.load(std::memory_order_relaxed);
  if (fc && fc != bs.pre_c) return 0;
  uint32_t best = 0;
  for (uint32_t k = 2; k <= 256; k <<= 1) {
    MsmShape sh = make_shape(1024, FpParams<CurveT<CID>::SF>::BITS, 0, bs.pre_c, k);
    if (!partition_supported(sh, true)) break;
    best = k;
  }
  return best;
}
template <int CID>
static void run_msm_batch(Ctx& c, const BaseSet& bs, size_t offset, const BatchItem* items, size_t k, const MsmCall& shared,
                          XYZZ<CurveT<CID>::BF>* results) {
  using C = CurveT<CID>;
  constexpr int BF = C::BF, SF = C::SF;
  require(bs.pre_W && k >= 1 && k <= 256 && !shared.u64_mode && !shared.gather_host && !shared.all_ones, NMX_E_ARG,
          "fused batch: unsupported call");
  const uint32_t sbits = FpParams<SF>::BITS;
  std::vector<uint32_t> off(k + 1), desc;
  std::vector<const uint32_t*> ptrs(k);
  uint64_t total_n = 0, lens_hash = 0x243f6a8885a308d3ull;
  for (size_t j = 0; j < k; j++) {
    off[j] = (uint32_t)total_n;
    total_n += items[j].n;
    lens_hash = (lens_hash ^ items[j].n) * 0x9e3779b97f4a7c15ull;
    lens_hash ^= lens_hash >> 29;
  }
  for (size_t j = 0; j < k; j++) results[j] = XYZZ<BF>::identity();
  if (total_n == 0) return;  // msm.rs:228, every vector
  require(total_n * bs.pre_W < 0xfff00000ull, NMX_E_TOO_LARGE, "n * windows must be < 2^32");
  off[k] = (uint32_t)total_n;

  MsmArgs a;
  a.scalars = nullptr;
  a.bases = bs.d;
  a.n = (uint32_t)total_n;
  a.scalars_mont = shared.scalars_mont ? 1u : 0u;
  a.u64_bits = 0;
  a.force_c = 0;
  a.force_lmax = G.force_lmax;
  a.force_fold_t = G.force_fold_t;
  a.pre_stride = (uint32_t)bs.n;
  a.pre_offset = (uint32_t)offset;
  a.pre_c = bs.pre_c;
  a.bases_clean = bs.any_identity ? 0u : 1u;
  a.no_partition = G.no_partition;
  a.seg_min_total = G.seg_min_total;
  a.seg_min_len = G.seg_min_len;
  a.seg_heavy_above = G.seg_heavy_above;
  a.big_slice = G.big_slice.load(std::memory_order_relaxed);
  a.hist_grid = G.hist_grid;
  a.hist_bs = G.hist_bs;
  {
    const uint32_t pf = G.accum_prefetch.load(std::memory_order_relaxed);
    a.accum_prefetch = pf ? pf : 1u;
  }
  a.batch_k = (uint32_t)k;
  c.wsum.resize(260);
  XYZZW* wsum = c.wsum.data();
  uint32_t err = 0;
  const bool prof = G.profiling.load(std::memory_order_relaxed);
  const uint64_t shape_key =
      (shape_hash(a, shared, 32, (uint32_t)CID, sbits, DeviceBackend(c, true, false).template seg_lanes<BF>((size_t)a.n * 16)) ^ lens_hash ^ (k << 48)) | 1u;
  for (int attempt = 0; attempt < 2; attempt++) try {
  for (int pass = (c.shape_key == shape_key && c.shape_bytes <= c.cap) ? 1 : 0; pass < 2; pass++) {
    DeviceBackend be(c, pass == 0, prof);
    // offsets and vector pointers in ONE allocation and one copy: [k + 1 offsets, padded to 8 bytes][k pointers]
    const size_t off_words = (k + 2) & ~(size_t)1;
    uint32_t* d_off = be.alloc<uint32_t>(off_words + 2 * k);
    const uint32_t** d_ptr = (const uint32_t**)(d_off + off_words);
    for (size_t j = 0; j < k; j++) {
      if (shared.scalars_device) {
        ptrs[j] = (const uint32_t*)items[j].scalars;
      } else {
        uint32_t* d_s = be.alloc<uint32_t>(items[j].n * 8);
        ptrs[j] = d_s;
        if (pass == 1 && items[j].n)
          HIPCHK(hipMemcpyAsync(d_s, items[j].scalars, items[j].n * 32, hipMemcpyHostToDevice, c.stream));
      }
    }
    if (pass == 1) {  // (a pageable source: the copy is staged before the call returns; `desc` outlives the sync)
      desc.assign(off_words + 2 * k, 0);
      memcpy(desc.data(), off.data(), (k + 1) * 4);
      memcpy(desc.data() + off_words, ptrs.data(), k * sizeof(void*));
      HIPCHK(hipMemcpyAsync(d_off, desc.data(), desc.size() * 4, hipMemcpyHostToDevice, c.stream));
    }
    a.batch_off = d_off;
    a.batch_vec = d_ptr;
    msm_pipeline<DeviceBackend, BF, SF>(be, a, sbits, wsum, &err);
    if (pass == 0) {
      arena_reserve(c, be.used);
      c.shape_key = shape_key;
      c.shape_bytes = be.used;
    } else if (prof) {
      float st[kMaxMarks];
      int ns = 0;
      for (int i = 0; i + 1 < be.nmarks; i++) {
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, c.ev[i], c.ev[i + 1]));
        st[ns++] = ms;
      }
      prof_store(st, ns);
    }
  }
  break;
  } catch (const Fail& f) {
    if (attempt == 1 || f.msg != "workspace arena overflow") throw;
    (void)hipStreamSynchronize(c.stream);
    c.shape_key = 0;
  }
  require(!(err & ERR_SCALAR_RANGE), NMX_E_SCALAR_RANGE, "scalar >= field modulus");
  for (size_t j = 0; j < k; j++) results[j] = XYZZ<BF>::load(wsum[j]);
}
template <int CID>
static void msm_key_entry(const BaseSet& bs, size_t offset, size_t n, const MsmCall& mc, uint32_t flags,
                          uint8_t* out, uint8_t* is_inf, Ctx& c) {
  auto r = run_msm_key<CID>(c, bs, offset, n, mc);
  write_result<CID>(r, flags, out, is_inf);
}


// ---------------------------------------------------------------------------------------------------
// type-erased table
// ---------------------------------------------------------------------------------------------------
template <int CID> struct CurveImpl {
  using C = CurveT<CID>;
  static constexpr int BF = C::BF, SF = C::SF;

  static void msm_key(Ctx& c, const BaseSet& bs, size_t offset, size_t n, const MsmCall& mc, uint32_t flags,
                      uint8_t* out, uint8_t* inf) {
    msm_key_entry<CID>(bs, offset, n, mc, flags, out, inf, c);
  }
  static void msm_key_batch(Ctx& c, const BaseSet& bs, size_t offset, const BatchItem* items, size_t k,
                            const MsmCall& shared, uint32_t flags, uint8_t* out, uint8_t* inf) {
    std::vector<XYZZ<BF>> r(k, XYZZ<BF>::identity());
    run_msm_batch<CID>(c, bs, offset, items, k, shared, r.data());
    for (size_t j = 0; j < k; j++) write_result<CID>(r[j], flags, out + 64 * j, inf ? inf + j : nullptr);
  }
  static uint32_t batch_limit(const BaseSet& bs) { return batch_limit_for<CID>(bs); }
  static size_t table_bytes(size_t n) {  // HBM a key of n points takes with its window tables
    uint32_t c = 0, W = 0;
    table_shape<CID>(n, NMX_BASES_PRECOMPUTE, &c, &W);
    return n * 64 * (size_t)(W ? W : 1);
  }
  // h * r on the host (~380 point operations, 0.1-0.2 ms); identity when r == 0
  static XYZZ<BF> blind_point(const void* h_xy64, const void* r, uint32_t flags) {
    uint32_t rw[8];
    memcpy(rw, r, 32);
    require(Fp<SF>::words_lt_p(rw), NMX_E_SCALAR_RANGE, "blinding scalar >= field modulus");
    if (flags & NMX_SCALARS_MONT) Fp<SF>::from_words(rw).mont256_to_canonical().to_words(rw);
    uint32_t any = 0;
    for (int i = 0; i < 8; i++) any |= rw[i];
    if (!any) return XYZZ<BF>::identity();
    Affine<BF> h;
    h.x = fp_from_bytes<BF>((const uint8_t*)h_xy64);
    h.y = fp_from_bytes<BF>((const uint8_t*)h_xy64 + 32);
    if (!h.is_identity()) {
      const bool m = (flags & NMX_BASES_MONT) != 0;
      h.x = (m ? h.x.mont256_to_internal() : h.x.to_internal()).canon();
      h.y = (m ? h.y.mont256_to_internal() : h.y.to_internal()).canon();
    }
    return scalar_mul<BF>(XYZZ<BF>::from_affine(h), rw);
  }
  static void blind_term(const void* h_xy64, const void* r, uint32_t flags, uint8_t* out128) {
    XYZZW w;
    blind_point(h_xy64, r, flags).store(w);
    memcpy(out128, w.w, 128);
  }
  static void commit(Ctx& c, const BaseSet& bs, size_t n, const MsmCall& mc, const void* h_xy64, const void* r,
                     uint32_t flags, uint8_t* out, uint8_t* inf) {
    // the blinding term is computed on a second host thread while the GPU runs the MSM, so a blinded commit costs what
    // an unblinded one does; the range check of r happens before anything is launched
    uint32_t rw[8];
    memcpy(rw, r, 32);
    require(Fp<SF>::words_lt_p(rw), NMX_E_SCALAR_RANGE, "blinding scalar >= field modulus");
    uint32_t any = 0;
    for (int i = 0; i < 8; i++) any |= rw[i];
    PoolFuture<XYZZ<BF>> hr;
    if (any) {
      std::array<uint8_t, 64> hb;
      std::array<uint8_t, 32> rb;
      memcpy(hb.data(), h_xy64, 64);
      memcpy(rb.data(), r, 32);
      hr = PoolFuture<XYZZ<BF>>([hb, rb, flags] { return blind_point(hb.data(), rb.data(), flags); });
    }
    auto acc = run_msm_key<CID>(c, bs, 0, n, mc);  // a failure here unwinds through hr's destructor, which waits
    if (any) acc.add(hr.get());
    write_result<CID>(acc, flags, out, inf);
  }
  static void commit_batch(Ctx& c, const BaseSet& bs, const BatchItem* items, size_t k, const MsmCall& shared, const void* h_xy64,
                           const uint8_t* rs32, const std::function<void(size_t, uint8_t*)>* late, uint32_t flags, uint8_t* out,
                           uint8_t* inf) {
    std::vector<PoolFuture<XYZZ<BF>>> hr(k);
    for (size_t j = 0; j < k; j++) {  // every range check before anything is launched
      require(items[j].n <= bs.n, NMX_E_HANDLE, "ck shorter than v");
      if (late) continue;
      uint32_t rw[8];
      memcpy(rw, rs32 + 32 * j, 32);
      require(Fp<SF>::words_lt_p(rw), NMX_E_SCALAR_RANGE, "blinding scalar >= field modulus");
    }
    std::array<uint8_t, 64> hb;
    memcpy(hb.data(), h_xy64, 64);
    for (size_t j = 0; j < k; j++) {
      if (late) {
        const std::function<void(size_t, uint8_t*)> get = *late;
        hr[j] = PoolFuture<XYZZ<BF>>([hb, get, j, flags] {
          uint8_t rb[32];
          get(j, rb);
          return blind_point(hb.data(), rb, flags);
        });
      } else {
        std::array<uint8_t, 32> rb;
        memcpy(rb.data(), rs32 + 32 * j, 32);
        hr[j] = PoolFuture<XYZZ<BF>>([hb, rb, flags] { return blind_point(hb.data(), rb.data(), flags); });
      }
    }
    std::vector<XYZZ<BF>> r(k, XYZZ<BF>::identity());
    if (k >= 2 && batch_limit_for<CID>(bs) >= k) {
      run_msm_batch<CID>(c, bs, 0, items, k, shared, r.data());  // a failure unwinds through hr's destructors, which wait
    } else {
      for (size_t j = 0; j < k; j++) {
        MsmCall mc = shared;
        mc.scalars = items[j].scalars;
        r[j] = run_msm_key<CID>(c, bs, 0, items[j].n, mc);
      }
    }
    for (size_t j = 0; j < k; j++) {
      r[j].add(hr[j].get());  // (h * 0 is the identity)
      write_result<CID>(r[j], flags, out + 64 * j, inf ? inf + j : nullptr);
    }
  }
  static void upload(Ctx& c, BaseSet& bs, const void* src, uint32_t flags, const BaseFill* fill) {
    upload_bases<CID>(c, bs, src, flags, fill);
  }
  static bool check_point_host(const uint8_t* xy64, uint32_t flags, uint8_t* out) {
    uint32_t w[16], err = 0;
    memcpy(w, xy64, 64);
    ValidateFn<CID> f{w, (flags & NMX_BASES_MONT) ? 1u : 0u, &err};
    f(0);
    if (err) return false;
    for (int k = 0; k < 2; k++) {
      Fp<BF> v = Fp<BF>::from_words(w + 8 * k);
      fp_to_bytes((flags & NMX_BASES_MONT) ? v.mont256_to_canonical() : v.canon(), out + 32 * k);
    }
    return true;
  }
  static void generate(Ctx& c, BaseSet& bs, uint64_t k0, uint32_t flags) {
    void* d = nullptr;
    const size_t n = bs.n;
    table_shape<CID>(n, flags, &bs.pre_c, &bs.pre_W);
    apply_table_limit(n, &bs.pre_c, &bs.pre_W);
    if (n) HIPCHK(hipMalloc(&d, n * 64 * (bs.pre_W ? bs.pre_W : 1)));
    try {
      DeviceBackend be(c, false, false);
      GenFn<CID> f{(AffineW*)d, k0};
      be.launch(f, (uint32_t)n);
      build_tables<CID>(c, d, n, bs.pre_c, bs.pre_W);
      bs.any_identity = scan_identity(c, d, n);  // (k0 + i) * G is the identity only if k0 + i = 0 mod r; also syncs
    } catch (...) {
      if (d) (void)hipFree(d);
      throw;
    }
    bs.d = d;
  }
  static void internal_to_canonical(uint8_t* e, size_t count) {
    for (size_t i = 0; i < count; i++) {
      Fp<BF> f = fp_from_bytes<BF>(e + 32 * i);
      fp_to_bytes(f.to_canonical(), e + 32 * i);
    }
  }
  static void point_sum(const uint8_t* partials128, size_t count, uint32_t flags, uint8_t* out, uint8_t* inf) {
    XYZZ<BF> acc = XYZZ<BF>::identity();
    for (size_t i = 0; i < count; i++) {
      XYZZW w;
      memcpy(w.w, partials128 + 128 * i, 128);
      acc.add(XYZZ<BF>::load(w));
    }
    write_result<CID>(acc, flags, out, inf);
  }
  // nmx_check_layout: raw bytes of the standard generator and of Scalar::from(value) must be x * 2^256 mod p limbs
  static bool check_layout(const uint8_t* gen64, const uint8_t* s32, uint64_t value) {
    uint32_t w[8];
    for (int k = 0; k < 2; k++) {
      memcpy(w, gen64 + 32 * k, 32);
      if (!Fp<BF>::words_lt_p(w)) return false;
      uint32_t got[8];
      Fp<BF>::from_words(w).mont256_to_canonical().to_words(got);
      if (memcmp(got, k ? C::GY : C::GX, 32) != 0) return false;
    }
    memcpy(w, s32, 32);
    if (!Fp<SF>::words_lt_p(w)) return false;
    uint32_t got[8], want[8] = {(uint32_t)value, (uint32_t)(value >> 32), 0, 0, 0, 0, 0, 0};
    Fp<SF>::from_words(w).mont256_to_canonical().to_words(got);
    return memcmp(got, want, 32) == 0;
  }
  static CurveOps ops() {
    return CurveOps{&msm_key, &msm_key_batch, &batch_limit, &commit, &upload, &check_point_host, FpParams<BF>::PW,
                    &generate, &internal_to_canonical, &point_sum, &blind_term, &check_layout, &table_bytes, &commit_batch, SF};
  }
};

}  // namespace nmx
