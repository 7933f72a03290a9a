// capi.hip -- the C ABI of include/nova_mi355x.h: global state, context pool, key registry, dispatch to the
// per-curve operation tables (curve_*.hip).  No group arithmetic and no CPU fallback in this file.
#include <algorithm>
#include <array>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <list>
#include <thread>

#include "runtime.hpp"
#include "combine.hpp"

namespace nmx {

// Process-lifetime singletons, deliberately never destroyed: keys are reference-counted and free their HBM in their
// destructor, and running those destructors during static destruction at process exit would call into a HIP runtime
// that may already be gone.  nmx_shutdown releases everything explicitly; a process that simply exits lets the driver
// reclaim the device memory.
Global& G = *new Global;
static std::atomic<uint64_t> g_stats[NMX_STAT_COUNT];
static inline void stat_add(int k, uint64_t v = 1) { g_stats[k].fetch_add(v, std::memory_order_relaxed); }

static inline int hip_device_of(int logical) {  // hip_dev only grows, within reserved capacity: safe to read unlocked
  return (size_t)logical < G.hip_dev.size() ? G.hip_dev[(size_t)logical] : G.device;
}
BaseSet::~BaseSet() {
  if (d && owns) {
    // the last reference can be dropped on any thread in the middle of its own call (an eviction, an unregister): free on the
    // key's device, then give the thread its device back
    int prev = -1;
    (void)hipGetDevice(&prev);
    (void)hipSetDevice(hip_device_of(dev));
    (void)hipFree(d);
    if (prev >= 0) (void)hipSetDevice(prev);
  }
}
Global::SparseSet::~SparseSet() {
  if (indptr || indices || data) (void)hipSetDevice(G.device);
  if (indptr) (void)hipFree(indptr);
  if (indices) (void)hipFree(indices);
  if (data) (void)hipFree(data);
}
Global::SparseSet::Transposed::~Transposed() {
  (void)hipSetDevice(hip_device_of(dev));  // the device transposed_of built it on
  for (uint32_t* q : {vptr, indices, data, vout, hrow, hstart})
    if (q) (void)hipFree(q);
}
void note_table_fallback() { stat_add(NMX_STAT_TABLE_FALLBACKS); }
void note_scan_timeout() { stat_add(NMX_STAT_SCAN_TIMEOUTS); }
// What one dependent few-wave launch costs on this box: 16 one-wave kernels chained on a stream between two events, best of
// three, once per process.  The fast boxes of the pool chain such launches back to back; BENCH_r02's box paid ~10 us each.
__global__ void k_gap_probe(uint32_t* p) {
  if (p) *p = 1;
}
int32_t launch_gap_ns(hipStream_t stream) {
  static std::once_flag once;
  std::call_once(once, [&] {
    int32_t best = 0x7fffffff;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess) {
      for (int rep = 0; rep < 4; rep++) {  // rep 0 warms the code object up
        (void)hipEventRecord(e0, stream);
        for (int i = 0; i < 16; i++) hipLaunchKernelGGL(k_gap_probe, dim3(1), dim3(64), 0, stream, (uint32_t*)nullptr);
        (void)hipEventRecord(e1, stream);
        float ms = 0;
        if (hipEventSynchronize(e1) == hipSuccess && hipEventElapsedTime(&ms, e0, e1) == hipSuccess && rep > 0) {
          const int32_t ns = (int32_t)(ms * 1e6f / 16.0f);
          if (ns < best) best = ns;
        }
      }
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipGetLastError();
    G.launch_gap_ns.store(best == 0x7fffffff ? 0 : best);
    g_stats[NMX_STAT_LAUNCH_GAP_NS].store((uint64_t)G.launch_gap_ns.load());
  });
  return G.launch_gap_ns.load(std::memory_order_relaxed);
}
static thread_local std::string t_err;
static thread_local float t_prof[kMaxMarks];
static thread_local int t_prof_n = 0;
static RcclCombine& RC = *new RcclCombine;  // never destroyed (see G); nmx_shutdown releases its communicators
// Record of the calling thread's last MSM over a multi-device key (nmx_profile_last_sharded): where every shard's scalars came
// from, its stage times (profiling on) and what the combine step cost.
struct ShardRec {
  int dev = 0, branch = 0, nst = 0;
  float ms[kMaxMarks] = {0};
};
static thread_local std::vector<ShardRec> t_shards;
static thread_local float t_combine_ms = 0;
static thread_local int t_rccl_ranks = 0;

void prof_store(const float* ms, int n) {
  t_prof_n = n < kMaxMarks ? n : kMaxMarks;
  for (int i = 0; i < t_prof_n; i++) t_prof[i] = ms[i];
}
void prof_add_tail(float ms) {
  if (t_prof_n > 0) t_prof[t_prof_n - 1] += ms;
}

void arena_reserve(Ctx& c, size_t bytes) {
  if (bytes <= c.cap) return;
  if (c.arena) HIPCHK(hipFree(c.arena));
  c.arena = nullptr;
  c.cap = 0;
  size_t want = bytes + bytes / 8 + (1u << 20);
  HIPCHK(hipMalloc((void**)&c.arena, want));
  c.cap = want;
}

void aux_reserve(Ctx& c, size_t bytes) {
  if (bytes <= c.aux_cap) return;
  if (c.aux) HIPCHK(hipFree(c.aux));
  c.aux = nullptr;
  c.aux_cap = 0;
  HIPCHK(hipMalloc((void**)&c.aux, bytes + (1u << 16)));
  c.aux_cap = bytes + (1u << 16);
}

static void cache_init_defaults();  // slice-cache budget from the device's HBM size / the environment (below)
static void ensure_init() {
  std::lock_guard<std::mutex> lk(G.mu);
  if (G.inited) return;
  int cnt = 0;
  hipError_t e = hipGetDeviceCount(&cnt);
  if (e != hipSuccess || cnt <= 0)
    throw Fail{NMX_E_NO_DEVICE, "no HIP device visible (libnova_mi355x has no CPU fallback)"};
  int dev = G.device;
  if (dev < 0) {
    const char* lr = getenv("LOCAL_RANK");
    dev = lr ? atoi(lr) % cnt : 0;
  }
  if (dev >= cnt) throw Fail{NMX_E_ARG, "device index out of range"};
  G.device = dev;
  G.hip_dev.reserve(Global::kMaxDevices);
  G.hip_dev.assign(1, dev);
  G.free_ctx.assign(Global::kMaxDevices, {});
  G.ndev_active.store(1);
  if (const char* t = getenv("NMX_SHARD_MIN_N")) G.shard_min_n.store((size_t)atoll(t));
  if (const char* t = getenv("NMX_MAX_TABLE_MIB")) G.max_table_bytes.store((size_t)atoll(t) << 20);
  if (const char* t = getenv("NMX_TUNE_LMAX")) G.force_lmax = (uint32_t)atoi(t);
  if (const char* t = getenv("NMX_TUNE_PRECOMP_MIN_N")) G.precomp_min_n = (size_t)atoll(t);
  if (const char* t = getenv("NMX_TUNE_FOLD_T")) {
    // the middle fold pass takes 2..63 lanes per bucket (1 = no middle pass); anything else would make the last
    // pass re-add positions the 64-lane pass already folded -- a tuning knob must not change results
    const int v = atoi(t);
    G.force_fold_t = v < 1 ? 0u : v > 63 ? 63u : (uint32_t)v;
  }
  if (const char* t = getenv("NMX_TUNE_NO_QUAD_ACCUM")) G.no_quad_accum = (uint32_t)atoi(t);
  if (const char* t = getenv("NMX_TUNE_NO_PARTITION")) G.no_partition = (uint32_t)atoi(t);
  if (const char* t = getenv("NMX_TUNE_SEG_MIN_TOTAL")) G.seg_min_total = (uint32_t)strtoul(t, nullptr, 0);
  if (const char* t = getenv("NMX_TUNE_SEG_MIN_LEN")) G.seg_min_len = (uint32_t)atoi(t) ? (uint32_t)atoi(t) : 1u;
  if (const char* t = getenv("NMX_TUNE_SEG_LANES")) G.seg_lanes_override = (uint32_t)atoi(t);
  if (const char* t = getenv("NMX_TUNE_NO_QUAD_FINAL")) G.no_quad_final = (uint32_t)atoi(t);
  if (const char* t = getenv("NMX_TUNE_SMALL_BLOCKS")) G.small_blocks = (uint32_t)atoi(t);
  if (const char* t = getenv("NMX_TUNE_QUAD_FINAL_BELOW")) G.quad_final_below = (uint32_t)atoi(t);
  if (const char* t = getenv("NMX_TUNE_NO_BATCH_FUSE")) G.no_batch_fuse = (uint32_t)atoi(t);
  if (const char* t = getenv("NMX_TUNE_PREFIX_TABLES")) G.prefix_tables = (uint32_t)atoi(t);
  if (const char* t = getenv("NMX_TUNE_NO_TREE_FUSE")) G.no_tree_fuse = (uint32_t)atoi(t);
  if (const char* t = getenv("NMX_TUNE_TREE_THREADS")) G.tree_threads = (uint32_t)atoi(t);
  if (const char* t = getenv("NMX_TUNE_BIG_SLICE")) G.big_slice = (uint32_t)atoi(t);
  if (const char* t = getenv("NMX_TUNE_BIG_THREADS")) G.big_threads = (uint32_t)atoi(t);
  if (const char* t = getenv("NMX_TUNE_HIST_GRID")) G.hist_grid = (uint32_t)atoi(t);
  if (const char* t = getenv("NMX_TUNE_HIST_BS")) G.hist_bs = (uint32_t)atoi(t);
  if (const char* t = getenv("NMX_SYNC_SPIN_US")) G.sync_spin_us = (uint32_t)atoi(t);
  if (const char* t = getenv("NMX_HOST_SPLIT")) {  // same range as option host_split: at most 16 pieces, 255 = by size
    const int v = atoi(t);
    G.host_split = v == 255 ? 255u : (uint32_t)(v < 0 ? 0 : (v > 16 ? 16 : v));
  }
  if (const char* t = getenv("NMX_SC_POLL_US")) G.sc_poll_us = (uint32_t)atoi(t);
  if (const char* t = getenv("NMX_SC_HOST_TAIL")) G.sc_host_tail = (uint32_t)atoi(t);
  if (const char* t = getenv("NMX_SC_FUSED_SUM")) G.sc_fused_sum = (uint32_t)atoi(t);
  if (const char* t = getenv("NMX_SC_SIDE_STREAMS")) G.sc_side_streams = (uint32_t)atoi(t);
  if (const char* t = getenv("NMX_SC_QUAD")) G.sc_quad = (uint32_t)atoi(t);
  if (const char* t = getenv("NMX_SC_PRELAUNCH")) G.sc_prelaunch = (uint32_t)atoi(t);
  if (const char* t = getenv("NMX_SC_HOST_PARTS")) G.sc_host_parts = (uint32_t)atoi(t);
  if (const char* t = getenv("NMX_SC_RESIDENT")) G.sc_resident = atoi(t) ? 1u : 0u;
  if (const char* t = getenv("NMX_TUNE_HORNER_TOP")) G.horner_top = (uint32_t)atoi(t);
  if (const char* t = getenv("NMX_TUNE_HORNER_SUB")) G.horner_sub = (uint32_t)atoi(t);
  if (const char* t = getenv("NMX_TUNE_HORNER_ORDER")) G.horner_order = atoi(t) ? 1u : 0u;
  if (const char* t = getenv("NMX_TUNE_EQ_MAX_BLOCKS")) G.eq_max_blocks = (uint32_t)atoi(t);
  if (const char* t = getenv("NMX_TUNE_SEG_HEAVY_ABOVE")) G.seg_heavy_above = (uint32_t)atoi(t);
  if (const char* t = getenv("NMX_TUNE_ACCUM_PF")) G.accum_prefetch = (uint32_t)atoi(t);
  HIPCHK(hipSetDevice(dev));
  cache_init_defaults();
  if (const char* t = getenv("NMX_DEVICES")) {  // same as nmx_init_devices(k, 0), for hosts that cannot call it
    const int k = atoi(t);
    if (k > cnt || k < 0) throw Fail{NMX_E_NO_DEVICE, "NMX_DEVICES asks for more devices than are visible"};
    for (int i = 1; i < (k ? k : cnt); i++) G.hip_dev.push_back((dev + i) % cnt);
    G.ndev_active.store((uint32_t)(k ? k : cnt));
  }
  G.inited = true;
}

// Stream-ordered calls (NMX_ASYNC): the element-wise field kernels and SpMV on HBM-resident vectors may return as soon as their
// kernel is enqueued -- between the MSMs of a prove_step the reference's host does nothing with W, E, T but hand them to the
// next provider call (src/r1cs/mod.rs:590-622, 1044-1107), so the 12 us wake-up of a blocking wait per call bought nothing.
// Ordering: a host thread's calls normally lease the same context (the pool is LIFO) and so share a stream; when they do not,
// the lease makes the new context's stream wait for the event recorded behind the thread's last asynchronous call.  Every
// synchronous call (all MSMs, all reductions, anything with a host operand) therefore still returns with everything the thread
// enqueued before it complete.  Vectors handed to ANOTHER host thread need nmx_sync() first.
static thread_local Ctx* t_async_ctx = nullptr;
static thread_local uint64_t t_async_epoch = 0;
static std::atomic<uint64_t> g_ctx_epoch{1};  // bumped by nmx_shutdown: contexts remembered by other threads are gone
static inline Ctx* async_pending() { return (t_async_ctx && t_async_epoch == g_ctx_epoch.load(std::memory_order_acquire)) ? t_async_ctx : nullptr; }
void async_mark(Ctx& c) {
  if (!c.async_ev) HIPCHK(hipEventCreateWithFlags(&c.async_ev, hipEventDisableTiming));
  HIPCHK(hipEventRecord(c.async_ev, c.stream));
  t_async_ctx = &c;
  t_async_epoch = g_ctx_epoch.load(std::memory_order_acquire);
}

// One context per in-flight call: concurrent callers (rayon workers on the reference side) never share a stream
// or a workspace, so a small MSM does not queue behind a 2^20 one.
// The lease also makes the context's device current on the calling thread (hipSetDevice is per thread).
struct CtxLease {
  Ctx* c;
  explicit CtxLease(int dev = 0) {
    ensure_init();
    require(dev >= 0 && (size_t)dev < G.hip_dev.size(), NMX_E_ARG, "logical device out of range");
    HIPCHK(hipSetDevice(hip_device_of(dev)));
    {
      std::lock_guard<std::mutex> lk(G.mu);
      auto& pool = G.free_ctx[(size_t)dev];
      if (!pool.empty()) {
        c = pool.back();
        pool.pop_back();
      } else {
        c = nullptr;
      }
    }
    if (!c) {
      c = new Ctx();
      c->dev = dev;
      HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
      std::lock_guard<std::mutex> lk(G.mu);
      G.all_ctx.push_back(c);
    }
    // behind this thread's last asynchronous call (same context: the stream already orders it)
    if (Ctx* p = async_pending(); p && p != c && p->async_ev) {
      const hipError_t e = hipStreamWaitEvent(c->stream, p->async_ev, 0);
      if (e != hipSuccess) {
        std::lock_guard<std::mutex> lk(G.mu);
        G.free_ctx[(size_t)c->dev].push_back(c);
        throw Fail{NMX_E_HIP, std::string("hipStreamWaitEvent: ") + hipGetErrorString(e)};
      }
    }
  }
  CtxLease(const CtxLease&) = delete;
  CtxLease& operator=(const CtxLease&) = delete;
  ~CtxLease() {
    std::lock_guard<std::mutex> lk(G.mu);
    G.free_ctx[(size_t)c->dev].push_back(c);
  }
};

static const CurveOps& ops(int curve) {
  switch (curve) {
    case NMX_BN254_G1: return curve_ops_bn254_g1();
    case NMX_GRUMPKIN: return curve_ops_grumpkin();
    case NMX_PALLAS: return curve_ops_pallas();
    case NMX_VESTA: return curve_ops_vesta();
  }
  throw Fail{NMX_E_ARG, "bad curve id"};
}

// A reference to the key: the caller's copy keeps the HBM allocation alive for the duration of its call even if
// another thread unregisters the handle meanwhile.
static BaseRef lookup(uint64_t h) {
  std::lock_guard<std::mutex> lk(G.mu);
  auto it = G.bases.find(h);
  if (it == G.bases.end()) throw Fail{NMX_E_HANDLE, "unknown base handle"};
  return it->second;
}
static uint64_t publish(std::shared_ptr<BaseSet> bs) {
  std::lock_guard<std::mutex> lk(G.mu);
  uint64_t h = G.next_handle++;
  G.bases[h] = std::move(bs);
  return h;
}
// key[offset, offset + n) inside the registered key?  Written so that offset + n cannot wrap (the Rust slice would
// have panicked; a C caller must get an error, not an out-of-bounds HBM read).
static inline bool slice_ok(const BaseSet& bs, size_t offset, size_t n) { return offset <= bs.n && n <= bs.n - offset; }
static_assert(kMaxMarks == NMX_PROF_STAGES, "nmx_profile_last_sharded's row length");
static MsmCall field_call(const void* scalars, uint32_t flags) {
  MsmCall m{scalars, (flags & (NMX_SCALARS_DEVICE | NMX_SCALARS_SHARDED)) != 0, (flags & NMX_SCALARS_MONT) != 0, 0, false};
  m.scalars_sharded = (flags & NMX_SCALARS_SHARDED) != 0;
  return m;
}

// ---------------------------------------------------------------------------------------------------
// Keys over several devices of ONE process.  The reference decomposes an MSM inside
// one address space -- `par_chunks` + `reduce(identity, +)`, /root/reference/src/provider/msm.rs:564-574,664-676 -- so
// the Rust host that binds this library is one process too: a key registered after nmx_init_devices(k) is cut into k
// contiguous shards (same rule as nova_amd/dist.py shard_range), shard i resident on logical device i with its own window
// tables, and every MSM over it fans out one host thread + stream per device touched, collects one 128-byte partial per
// shard and sums them on the host.  No bucket array ever crosses devices.
// ---------------------------------------------------------------------------------------------------
struct PartRange {
  size_t begin, n;
};
static inline PartRange shard_range(size_t n, uint32_t i, uint32_t k) {
  const size_t base = n / k, rem = n % k;
  return PartRange{(size_t)i * base + (i < rem ? i : rem), base + (i < rem ? 1u : 0u)};
}
WorkerPool& worker_pool() {
  static WorkerPool& wp = *new WorkerPool;  // leaked on purpose (see WorkerPool)
  return wp;
}
static WorkerPool& WP = worker_pool();

// fn(i) for i in [0, count): i = 0 on the calling thread, the others on pool workers when `parallel`.  Nothing escapes a
// worker; the first failure is rethrown here, after every part has finished.  The calling thread ends on the primary device.
template <class Fn> static void run_on_parts(size_t count, bool parallel, Fn&& fn) {
  std::mutex err_mu;
  bool failed = false;
  Fail first{0, ""};
  Ctx* const pend = async_pending();  // the caller's last asynchronous call: leases on the helper threads must wait for it too
  const uint64_t pend_epoch = t_async_epoch;
  const std::thread::id caller = std::this_thread::get_id();
  auto guarded_fn = [&](size_t i) {
    // helper threads borrow the caller's pending asynchronous mark for the duration of fn(i) and get their own state back;
    // on the CALLING thread nothing is restored: an asynchronous mark fn(i) records there must survive the call, or the
    // thread's next lease / nmx_sync would not wait for it
    struct Restore {
      bool on;
      Ctx* v;
      uint64_t e;
      ~Restore() {
        if (on) t_async_ctx = v, t_async_epoch = e;
      }
    } restore{std::this_thread::get_id() != caller, t_async_ctx, t_async_epoch};
    if (restore.on) t_async_ctx = pend, t_async_epoch = pend_epoch;
    try {
      fn(i);
    } catch (const Fail& f) {
      std::lock_guard<std::mutex> lk(err_mu);
      if (!failed) first = f;
      failed = true;
    } catch (const std::exception& e) {
      std::lock_guard<std::mutex> lk(err_mu);
      if (!failed) first = Fail{NMX_E_HIP, e.what()};
      failed = true;
    } catch (...) {
      std::lock_guard<std::mutex> lk(err_mu);
      if (!failed) first = Fail{NMX_E_HIP, "unknown exception in a device worker"};
      failed = true;
    }
  };
  if (!parallel || count <= 1) {
    for (size_t i = 0; i < count; i++) guarded_fn(i);
  } else {
    std::mutex done_mu;
    std::condition_variable done_cv;
    size_t pending = 0, started = 1;
    try {
      for (size_t i = 1; i < count; i++) {
        Worker* w = WP.acquire();
        {
          std::lock_guard<std::mutex> lk(done_mu);
          pending++;
        }
        try {
          WP.submit(w, [&, w, i] {
            guarded_fn(i);
            WP.release(w);
            std::lock_guard<std::mutex> lk(done_mu);  // held across the notify: the waiter cannot leave (and destroy done_cv) in between
            pending--;
            done_cv.notify_one();
          });
        } catch (...) {
          {
            std::lock_guard<std::mutex> lk(done_mu);
            pending--;
          }
          WP.release(w);
          throw;
        }
        started = i + 1;
      }
    } catch (const std::exception&) {  // std::system_error from thread creation: the rest runs on this thread
    }
    guarded_fn(0);
    for (size_t i = started; i < count; i++) guarded_fn(i);
    std::unique_lock<std::mutex> lk(done_mu);
    done_cv.wait(lk, [&] { return pending == 0; });
  }
  (void)hipSetDevice(G.device);
  if (failed) throw first;
}
// Narrower table sets over a key's first points (BaseSet::prefix, a chain).  `key` is a complete single-device key (a whole key,
// or shard 0 of a sharded one: the prefix of the whole key lives there).  Levels: 2^18 points (c = 16), 2^16 (c = 15), 2^13
// (c = 8); a level is added when it is at most half the key and at least two bits narrower than the key's own tables:
//   >= 2^22 points (c = 20) -> 2^18 -> 2^13        2^20 .. 2^21 (c = 17) -> 2^16 -> 2^13        2^14 .. 2^19 (c = 15 / 16) -> 2^13
// Single MSMs / commitments use the first level of a WIDE key only (2^15 buckets to reduce instead of 2^19); batches walk the
// whole chain with option prefix_tables = 2: every vector runs, fused with its peers, on the narrowest level that holds it
// (a fused run reduces sets x 2^(c-1) buckets: sixteen sets of a c = 17 key are 2^20 buckets for a handful of pairs).
// Out of memory: the key simply has no (further) prefix.
static void add_prefix_tables(BaseSet& key) {
  const uint32_t mode = G.prefix_tables.load(std::memory_order_relaxed);
  if (!mode || key.prefix || !key.d || !key.pre_W) return;
  static const struct { size_t n; uint32_t c; } kLevels[3] = {{(size_t)1 << 18, 16}, {(size_t)1 << 16, 15}, {(size_t)1 << 13, 8}};
  size_t want = 0;
  for (const auto& lv : kLevels)
    if (2 * lv.n <= key.n && lv.c + 2 <= key.pre_c && (mode >= 2 || key.pre_c >= 18)) {
      want = lv.n;
      break;
    }
  if (!want) return;
  const CurveOps& o = ops(key.curve);
  auto pre = std::make_shared<BaseSet>(key.curve, want);
  pre->dev = key.dev;
  try {
    CtxLease L(key.dev);
    o.upload(*L.c, *pre, key.d, NMX_BASES_DEVICE | NMX_BASES_INTERNAL | NMX_BASES_PRECOMPUTE, nullptr);
  } catch (const Fail& f) {
    (void)hipSetDevice(G.device);
    if (f.code == NMX_E_HIP && f.msg.find("out of memory") != std::string::npos) {
      stat_add(NMX_STAT_TABLE_FALLBACKS);
      return;
    }
    throw;
  }
  (void)hipSetDevice(G.device);
  if (!pre->pre_W) return;  // (a prefix that fell back to no tables is of no use)
  if (mode >= 2) add_prefix_tables(*pre);
  key.prefix = pre;
}
// the key object an MSM over key[offset, offset + n) of a single-device key should run on: the first prefix level of a wide key
static inline const BaseSet& prefix_or_key(const BaseSet& key, size_t offset, size_t n) {
  return (key.prefix && key.pre_c >= 18 && n && offset + n <= key.prefix->n) ? *key.prefix : key;
}
// A key of n points: whole on the primary device (mk runs on `c0`), or -- allow_shard, more than one active device, at
// least shard_min_n points -- sharded.  mk(ctx, part, begin) fills `part` (part.n points starting at point `begin` of the
// key) on ctx's device: an upload, a file read, a generator.
template <class MakePart>
static std::shared_ptr<BaseSet> build_key(Ctx& c0, int curve, size_t n, bool allow_shard, bool parallel, MakePart&& mk) {
  const uint32_t k = allow_shard ? G.ndev_active.load(std::memory_order_relaxed) : 1u;
  auto bs = std::make_shared<BaseSet>(curve, n);
  if (k <= 1 || n < G.shard_min_n.load(std::memory_order_relaxed) || n < k) {
    mk(c0, *bs, (size_t)0);
    add_prefix_tables(*bs);
    return bs;
  }
  bs->parts.resize(k);
  bs->part_begin.resize(k + 1);
  for (uint32_t i = 0; i < k; i++) {
    const PartRange r = shard_range(n, i, k);
    bs->parts[i] = std::make_shared<BaseSet>(curve, r.n);
    bs->parts[i]->dev = (int)i;
    bs->part_begin[i] = r.begin;
  }
  bs->part_begin[k] = n;
  run_on_parts(k, parallel, [&](size_t i) {
    CtxLease L((int)i);
    mk(*L.c, *bs->parts[i], bs->part_begin[i]);
  });
  add_prefix_tables(*bs->parts[0]);
  return bs;
}
// the pieces of key[offset, offset + n) by shard: (part, offset inside the part, count, offset inside the call)
struct PartJob {
  const BaseSet* part;
  size_t poff, cnt, goff;
};
static std::vector<PartJob> parts_of(const BaseSet& bs, size_t offset, size_t n) {
  std::vector<PartJob> jobs;
  for (size_t i = 0; i < bs.parts.size() && n; i++) {
    const size_t b = bs.part_begin[i], e = bs.part_begin[i + 1];
    const size_t lo = offset > b ? offset : b, hi = offset + n < e ? offset + n : e;
    if (lo < hi) jobs.push_back(PartJob{bs.parts[i].get(), lo - b, hi - lo, lo - offset});
  }
  return jobs;
}
// The combine step of a sharded call: `cnt` 128-byte partials, partial i computed on logical device devs[i].  With the shards
// on two or more GPUs: one slot per GPU (shards that share a GPU under NMX_DEVICES_OVERSUBSCRIBE are summed first; a GPU the
// call did not touch sends the identity), ONE ncclAllGather over xGMI, then the G-term point sum on the host from rank 0's
// copy -- k <= 8 additions of ~0.5 us each: a one-wave device kernel would pay ~2.6 us of latency per dependent addition
// (profiles/r03_msm_2p20/add_latency.txt).  Otherwise (one GPU, option combine = 1, RCCL not loadable): the host sum of
// the partials the shard workers already hold.
enum { NMX_COMBINE_AUTO = 0, NMX_COMBINE_HOST = 1, NMX_COMBINE_RCCL = 2 };
static void combine_partials(const CurveOps& o, const uint8_t* partials, const std::vector<int>& devs, uint32_t flags,
                             uint8_t* out, uint8_t* inf) {
  const auto t0 = std::chrono::steady_clock::now();
  const size_t cnt = devs.size();
  const uint32_t mode = G.combine_mode.load(std::memory_order_relaxed);
  int used_ranks = 0;
  std::vector<int> phys;  // distinct GPUs of the ACTIVE logical devices, in logical order: the communicator's ranks
  {
    const uint32_t nd = G.ndev_active.load(std::memory_order_relaxed);
    for (uint32_t i = 0; i < nd; i++) {
      const int h = hip_device_of((int)i);
      if (std::find(phys.begin(), phys.end(), h) == phys.end()) phys.push_back(h);
    }
  }
  // Inside one process the shard workers hand their partials to the calling thread in HOST memory (the window sums are finished
  // by the host tail of every MSM), so the host sum is the default: k x 128 bytes, <= 8 additions of ~0.5 us, nothing on a
  // stream.  RCCL here would be host -> device -> all-gather -> device -> host for the same bytes (2 copies + k stream syncs on
  // the critical path), so it is opt-in ("combine" = 2: the communicator, the collective and the xGMI route get exercised);
  // the deployment where the all-gather IS the exchange step is one process per GPU (nova_amd/dist.py, bench.py --gpus N under
  // torch.distributed.run), where no rank can see another rank's partial.
  const bool want = mode == NMX_COMBINE_RCCL;
  bool done = false;
  if (want && cnt >= 1) {
    std::lock_guard<std::mutex> lk(RC.mu);
    if (RC.ensure_locked(phys)) {
      try {
        const size_t R = phys.size();
        std::vector<uint8_t> slots(R * kCombineSlot, 0), gathered(R * kCombineSlot);  // zero bytes: ZZ = 0, the identity
        for (size_t r = 0; r < R; r++) {
          std::vector<uint8_t> mine;
          for (size_t i = 0; i < cnt; i++)
            if (hip_device_of(devs[i]) == phys[r]) mine.insert(mine.end(), partials + 128 * i, partials + 128 * i + 128);
          if (mine.size() == 128) memcpy(slots.data() + r * kCombineSlot, mine.data(), 128);
          else if (!mine.empty()) o.point_sum(mine.data(), mine.size() / 128, NMX_OUT_PARTIAL, slots.data() + r * kCombineSlot, nullptr);
        }
        RC.all_gather_locked(slots.data(), gathered.data());
        o.point_sum(gathered.data(), R, flags, out, inf);
        used_ranks = (int)R;
        done = true;
      } catch (const Fail&) {
        (void)hipGetLastError();
        if (mode == NMX_COMBINE_RCCL) {
          (void)hipSetDevice(G.device);
          throw;
        }
      }
    } else if (mode == NMX_COMBINE_RCCL) {
      (void)hipSetDevice(G.device);
      throw Fail{NMX_E_HIP, "option combine = 2 (RCCL required): " + RC.why};
    }
    (void)hipSetDevice(G.device);
  }
  if (!done) o.point_sum(partials, cnt, flags, out, inf);
  t_rccl_ranks = used_ranks;
  t_combine_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
}
// A piece of a raw NMX_SCALARS_SHARDED call carries no length: the caller cut the vector by a plan, and a plan made for the
// wrong key length (the classic: a key registered with its blinding point behind it has n + 1 points) would make the digit
// kernels read past the piece -- a GPU memory fault that ends the process.  Two driver queries per piece (~1 us each) turn
// that into an error: the piece must be device memory on the shard's GPU, and [p, p + bytes) must lie inside one allocation.
static void check_shard_piece(const void* p, size_t bytes, int logical_dev) {
  require(p != nullptr, NMX_E_ARG, "null shard pointer");
  hipPointerAttribute_t at;
  memset(&at, 0, sizeof at);
  if (hipPointerGetAttributes(&at, p) != hipSuccess || (at.type != hipMemoryTypeDevice && at.type != hipMemoryTypeManaged)) {
    (void)hipGetLastError();
    throw Fail{NMX_E_ARG, "NMX_SCALARS_SHARDED: a piece is not a device pointer"};
  }
  if (at.device != hip_device_of(logical_dev))
    throw Fail{NMX_E_ARG, "NMX_SCALARS_SHARDED: piece on HIP device " + std::to_string(at.device) + ", its shard lives on device " +
                              std::to_string(hip_device_of(logical_dev)) + " (cut the vector with nmx_bases_shard_plan)"};
  hipDeviceptr_t base = nullptr;
  size_t size = 0;
  if (hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)p) != hipSuccess) {
    (void)hipGetLastError();
    return;  // (not every allocator answers this query: the device check above stands)
  }
  const size_t inside = (size_t)((const char*)p - (const char*)base);
  require(inside <= size && bytes <= size - inside, NMX_E_ARG,
          "NMX_SCALARS_SHARDED: a piece is shorter than its shard's share of the call (cut the vector with nmx_bases_shard_plan: "
          "the plan must be made for the REGISTERED key length)");
}
// out = sum scalars[i] * key[offset + i] for any key: one device, or one partial per shard touched + the combine step
static void key_msm(Ctx& c0, const BaseSet& bs, size_t offset, size_t n, const MsmCall& mc, uint32_t flags, uint8_t* out,
                    uint8_t* inf) {
  const CurveOps& o = ops(bs.curve);
  t_shards.clear(), t_combine_ms = 0, t_rccl_ranks = 0;  // nmx_profile_last_sharded describes THIS call (no shards: single device)
  if (bs.parts.empty()) {
    if (mc.scalars_sharded) {  // one piece: the pointer array has one entry, on the key's device
      MsmCall m = mc;
      m.scalars = n ? ((const void* const*)mc.scalars)[0] : nullptr;
      m.scalars_device = true;
      m.scalars_sharded = false;
      if (n) check_shard_piece(m.scalars, n * (mc.u64_mode ? 8 : 32), bs.dev);
      o.msm_key(c0, prefix_or_key(bs, offset, n), offset, n, m, flags, out, inf);
      return;
    }
    // Host scalars of a large call (the trait's own form: `vartime_multiscalar_mul(&[Scalar], &ck[..n])`, bn256_grumpkin.rs:45-47):
    // the upload (32 MiB at 2^20: ~0.6 ms over PCIe) cannot overlap an MSM that needs every scalar before its buckets are
    // cut -- but it can overlap ANOTHER MSM.  The call is cut into `host_split` contiguous pieces (the reference's own
    // decomposition: chunks + reduce(identity, +), msm.rs:564-574), piece i's scalars cross PCIe while piece i - 1 computes;
    // the copies take turns (two copies at once would share the link and both finish late), each piece runs on its own
    // context and stream, the 128-byte partials are summed on the host.
    uint32_t split = G.host_split.load(std::memory_order_relaxed);
    // automatic (the default, 255): two pieces from 2^19 pairs, three from 2^20, four from 2^21 (measured: profiles/r05_msm_2p20/host_split.txt)
    if (split == 255u) split = n >= ((size_t)1 << 21) ? 4u : n >= ((size_t)1 << 20) ? 3u : 2u;
    if (split > 1 && mc.scalars && !mc.scalars_device && !mc.gather_host && !mc.all_ones && n >= G.host_split_min_n.load(std::memory_order_relaxed) &&
        n / split >= 4096) {
      const size_t sb = mc.u64_mode ? 8 : 32;
      std::vector<uint8_t> partials(128 * (size_t)split);
      std::mutex turn_mu;
      std::condition_variable turn_cv;
      uint32_t turn = 0;
      run_on_parts(split, true, [&](size_t i) {
        const PartRange r = shard_range(n, (uint32_t)i, split);
        CtxLease L(bs.dev);
        bool copied = false;
        auto pass_turn = [&] {  // whatever happens, the next piece must not wait for ever
          std::lock_guard<std::mutex> lk(turn_mu);
          if (turn == i) turn++;
          turn_cv.notify_all();
        };
        try {
          aux_reserve(*L.c, r.n * sb);
          {
            std::unique_lock<std::mutex> lk(turn_mu);
            turn_cv.wait(lk, [&] { return turn >= i; });
          }
          HIPCHK(hipMemcpyAsync(L.c->aux, (const char*)mc.scalars + r.begin * sb, r.n * sb, hipMemcpyHostToDevice, L.c->stream));
          HIPCHK(hipStreamSynchronize(L.c->stream));
          copied = true;
          pass_turn();
          MsmCall m = mc;
          m.scalars = L.c->aux;
          m.scalars_device = true;
          o.msm_key(*L.c, prefix_or_key(bs, offset + r.begin, r.n), offset + r.begin, r.n, m, (flags & ~(uint32_t)NMX_OUT_PARTIAL) | NMX_OUT_PARTIAL,
                    partials.data() + 128 * i, nullptr);
        } catch (...) {
          if (!copied) pass_turn();
          throw;
        }
      });
      o.point_sum(partials.data(), split, flags, out, inf);
      return;
    }
    o.msm_key(c0, mc.gather_host ? bs : prefix_or_key(bs, offset, n), offset, n, mc, flags, out, inf);
    return;
  }
  require(!mc.gather_host, NMX_E_ARG, "internal: sparse calls over a sharded key go through key_msm_sparse");
  const std::vector<PartJob> jobs = parts_of(bs, offset, n);
  std::vector<uint8_t> partials(128 * (jobs.size() ? jobs.size() : 1));
  std::vector<ShardRec> recs(jobs.size());
  const size_t sbytes = mc.u64_mode ? 8 : 32;
  const bool force_peer = G.force_peer_copy.load(std::memory_order_relaxed) != 0;
  if (mc.scalars_sharded)
    for (size_t i = 0; i < jobs.size(); i++)
      check_shard_piece(((const void* const*)mc.scalars)[i], jobs[i].cnt * sbytes, jobs[i].part->dev);
  run_on_parts(jobs.size(), true, [&](size_t i) {
    const PartJob& j = jobs[i];
    CtxLease L(j.part->dev);
    MsmCall m = mc;
    m.scalars_sharded = false;
    recs[i].dev = j.part->dev;
    recs[i].branch = NMX_BRANCH_NONE;
    if (mc.scalars_sharded) {
      // shard-resident operands (the reference chunks coefficients and bases together, msm.rs:564-574): piece i of the
      // scalars already lives on this shard's GPU -- nothing crosses xGMI or PCIe inside the call
      m.scalars = ((const void* const*)mc.scalars)[i];
      m.scalars_device = true;
      recs[i].branch = NMX_BRANCH_SHARD_RESIDENT;
    } else if (mc.scalars) {
      const char* src = (const char*)mc.scalars + j.goff * sbytes;
      if (mc.scalars_device && (hip_device_of(j.part->dev) != G.device || force_peer)) {
        // HBM-resident scalars live on the primary device: this shard's slice crosses xGMI once, peer to peer
        aux_reserve(*L.c, j.cnt * sbytes);
        HIPCHK(hipMemcpyPeerAsync(L.c->aux, hip_device_of(j.part->dev), src, G.device, j.cnt * sbytes, L.c->stream));
        m.scalars = L.c->aux;
        recs[i].branch = NMX_BRANCH_PEER_COPY;
      } else {
        m.scalars = src;
        recs[i].branch = mc.scalars_device ? NMX_BRANCH_LOCAL : NMX_BRANCH_HOST;
      }
    }
    o.msm_key(*L.c, prefix_or_key(*j.part, j.poff, j.cnt), j.poff, j.cnt, m, (flags & ~(uint32_t)NMX_OUT_PARTIAL) | NMX_OUT_PARTIAL,
              partials.data() + 128 * i, nullptr);
    if (G.profiling.load(std::memory_order_relaxed)) {  // this worker's stage times, handed to the calling thread below
      recs[i].nst = t_prof_n;
      for (int q = 0; q < t_prof_n; q++) recs[i].ms[q] = t_prof[q];
    }
  });
  stat_add(NMX_STAT_SHARDED_CALLS, 1);
  std::vector<int> devs(jobs.size());
  for (size_t i = 0; i < jobs.size(); i++) devs[i] = jobs[i].part->dev;
  combine_partials(o, partials.data(), devs, flags, out, inf);
  t_shards.swap(recs);
  // nmx_profile_last on the calling thread: the slowest shard stage by stage (the call waits for all of them)
  if (G.profiling.load(std::memory_order_relaxed) && !t_shards.empty()) {
    float mx[kMaxMarks] = {0};
    int ns = 0;
    for (const ShardRec& r : t_shards) {
      ns = r.nst > ns ? r.nst : ns;
      for (int q = 0; q < r.nst; q++) mx[q] = r.ms[q] > mx[q] ? r.ms[q] : mx[q];
    }
    prof_store(mx, ns);
  }
}
// sparse forms over any key: indices are positions in the whole key; scalars == nullptr: all ones
static void key_msm_sparse(Ctx& c0, const BaseSet& bs, const uint32_t* idx, const void* scalars, size_t k, uint32_t flags,
                           uint8_t* out, uint8_t* inf) {
  const CurveOps& o = ops(bs.curve);
  MsmCall mc = scalars ? field_call(scalars, flags) : MsmCall{nullptr, false, false, 1, true};
  mc.all_ones = scalars == nullptr;
  t_shards.clear(), t_combine_ms = 0, t_rccl_ranks = 0;
  if (bs.parts.empty()) {
    mc.gather_host = idx;
    o.msm_key(c0, bs, 0, k, mc, flags, out, inf);
    return;
  }
  require(!(flags & (NMX_SCALARS_DEVICE | NMX_SCALARS_SHARDED)), NMX_E_ARG, "sparse MSM over a multi-device key takes host scalars");
  const size_t np = bs.parts.size();
  std::vector<std::vector<uint32_t>> pidx(np);
  std::vector<std::vector<uint8_t>> psc(np);
  for (size_t t = 0; t < k; t++) {
    size_t p = std::upper_bound(bs.part_begin.begin(), bs.part_begin.end(), (size_t)idx[t]) - bs.part_begin.begin() - 1;
    pidx[p].push_back((uint32_t)(idx[t] - bs.part_begin[p]));
    if (scalars) psc[p].insert(psc[p].end(), (const uint8_t*)scalars + 32 * t, (const uint8_t*)scalars + 32 * t + 32);
  }
  std::vector<size_t> used;
  for (size_t p = 0; p < np; p++)
    if (!pidx[p].empty()) used.push_back(p);
  std::vector<uint8_t> partials(128 * (used.size() ? used.size() : 1));
  std::vector<ShardRec> recs(used.size());
  run_on_parts(used.size(), true, [&](size_t i) {
    const size_t p = used[i];
    CtxLease L(bs.parts[p]->dev);
    MsmCall m = mc;
    m.scalars = scalars ? psc[p].data() : nullptr;
    m.gather_host = pidx[p].data();
    recs[i].dev = bs.parts[p]->dev;
    recs[i].branch = scalars ? NMX_BRANCH_HOST : NMX_BRANCH_NONE;
    o.msm_key(*L.c, *bs.parts[p], 0, pidx[p].size(), m, (flags & ~(uint32_t)NMX_OUT_PARTIAL) | NMX_OUT_PARTIAL,
              partials.data() + 128 * i, nullptr);
    if (G.profiling.load(std::memory_order_relaxed)) {
      recs[i].nst = t_prof_n;
      for (int q = 0; q < t_prof_n; q++) recs[i].ms[q] = t_prof[q];
    }
  });
  t_shards.swap(recs);
  stat_add(NMX_STAT_SHARDED_CALLS, 1);
  std::vector<int> devs(used.size());
  for (size_t i = 0; i < used.size(); i++) devs[i] = bs.parts[used[i]]->dev;
  combine_partials(o, partials.data(), devs, flags, out, inf);
}

struct OrFn {  // OR of all u64 scalars -> bit length of the maximum
  const uint32_t* s;
  uint32_t* out;
  NMX_HD void operator()(uint32_t i) const {
    if (s[2 * (size_t)i]) nmx_atomic_or(out, s[2 * (size_t)i]);
    if (s[2 * (size_t)i + 1]) nmx_atomic_or(out + 1, s[2 * (size_t)i + 1]);
  }
};
static uint32_t resolve_u64_bits(Ctx& c, const uint64_t* s, size_t n, bool dev, uint32_t max_bits) {
  if (max_bits != NMX_BITS_AUTO) {
    require(max_bits <= 64, NMX_E_ARG, "max_num_bits must be <= 64");
    return max_bits;
  }
  if (n == 0) return 0;
  uint64_t orv = 0;
  if (!dev) {
    for (size_t i = 0; i < n; i++) orv |= s[i];
  } else {
    arena_reserve(c, 256);
    HIPCHK(hipMemsetAsync(c.arena, 0, 8, c.stream));
    DeviceBackend be(c, false, false);
    OrFn f{(const uint32_t*)s, (uint32_t*)c.arena};
    be.launch(f, (uint32_t)n);
    HIPCHK(hipMemcpyAsync(&orv, c.arena, 8, hipMemcpyDeviceToHost, c.stream));
    HIPCHK(hipStreamSynchronize(c.stream));
  }
  uint32_t b = 0;
  while (orv) {
    b++;
    orv >>= 1;
  }
  return b;  // num_bits(max) as msm.rs:456-462,473
}

template <class Fn> static int guarded(Fn&& fn) {
  try {
    fn();
    return NMX_OK;
  } catch (const Fail& f) {
    t_err = f.msg;
    return f.code;
  } catch (const std::exception& e) {
    t_err = e.what();
    return NMX_E_HIP;
  }
}


// ---------------------------------------------------------------------------------------------------
// Slice cache: device residency for the trait's slice-form calls (include/nova_mi355x.h, "Slice form").
// The reference passes `&ck.ck[..n]` (pedersen.rs:267, hyperkzg.rs:588) -- the address of element 0 of one long-lived
// Vec for every n -- so (curve, layout, address) names the key and sampled content fingerprints confirm it.
// ---------------------------------------------------------------------------------------------------
static inline uint64_t point_hash(const void* p64) {  // 64-bit multiply-xor hash of the 64 point bytes
  uint64_t w[8], h = 0x9e3779b97f4a7c15ull;
  memcpy(w, p64, 64);
  for (int i = 0; i < 8; i++) {
    h = (h ^ w[i]) * 0xff51afd7ed558ccdull;
    h ^= h >> 29;
  }
  return h;
}
// f(t) for t in [0, parts) on pool workers (part 0 on the calling thread), all joined before returning
template <class Fn> static void pool_for(size_t parts, Fn f) {
  std::vector<PoolFuture<bool>> futs;
  futs.reserve(parts);
  for (size_t t = 1; t < parts; t++) futs.emplace_back([f, t] { f(t); return true; });
  f(0);
  for (auto& x : futs) (void)x.get();
}
// hashes of points [lo, hi) of a host array into out[lo, hi), on up to 8 threads for long ranges (2^20 points: ~1 ms)
static void hash_points(const uint8_t* host, size_t lo, size_t hi, uint64_t* out) {
  const size_t n = hi - lo;
  const size_t nth = n < ((size_t)1 << 16) ? 1 : std::min<size_t>(8, n >> 15);
  pool_for(nth, [=](size_t t) {
    for (size_t i = lo + n * t / nth; i < lo + n * (t + 1) / nth; i++) out[i] = point_hash(host + 64 * i);
  });
}
// Verification of a hit, from the caller's bytes against the 64-bit hashes of ALL points recorded at upload.
//  * default (option cache_verify = 0): EVERY point of the slice on EVERY call -- the trait is a pure function of the
//    slice's contents (/root/reference/src/provider/traits.rs:79), so an in-place edit of a single point must change the
//    very next result.  Slices up to 2048 points are hashed at once; longer ones get a quick look (first, last, eight
//    moving probes: a freed-and-reused address fails here, before anything is launched) and the full pass runs on pool
//    workers WHILE the GPU computes the MSM (2^20 points: 64 MiB, ~1 ms on 8 threads, under a 2.2 ms call); the result
//    is handed out only after it has passed, otherwise the entry is dropped and the call repeated on a fresh upload.
//  * cache_verify = 1 (callers that register immutable keys): the quick look + a ROLLING WINDOW of max(4096, n/16) points
//    that continues where the previous call stopped -- an in-place edit is then caught within 16 calls, not at once.
static constexpr size_t kFullVerifyBelow = 2048, kSyncWindow = 4096, kWindowFraction = 16;
struct DeepCheck {  // the part of a hit's verification that may run beside the MSM; needs the caller's slice to stay valid
  std::shared_ptr<const std::vector<uint64_t>> ph;
  const uint8_t* slice = nullptr;
  size_t off = 0, start = 0, count = 0;  // points [start, start + count) of the slice against ph[off + ...]
  bool needed() const { return count != 0; }
  bool run_range(size_t a, size_t b) const {
    const uint64_t* h = ph->data() + off;
    for (size_t i = a; i < b; i++)
      if (point_hash(slice + 64 * i) != h[i]) return false;
    return true;
  }
  bool run() const { return run_range(start, start + count); }
  // on pool workers, up to 8 of them ACROSS ALL CALLERS (concurrent rayon callers hitting cached keys share the 8: a caller
  // that finds them taken verifies on one worker -- still every point, still joined before the result is handed out -- so
  // the check costs at most 8 host threads and their memory bandwidth, whatever the number of callers); every future must be
  // joined before the slice goes away
  static std::atomic<int>& inflight() {
    static std::atomic<int> v{0};
    return v;
  }
  std::vector<PoolFuture<bool>> run_async() const {
    size_t nth = std::max<size_t>(1, std::min<size_t>(8, count >> 15));
    const int busy = inflight().load(std::memory_order_relaxed);
    if ((size_t)busy + nth > 8) nth = busy >= 7 ? 1 : (size_t)(8 - busy);
    std::vector<PoolFuture<bool>> futs;
    futs.reserve(nth);
    const DeepCheck dc = *this;
    for (size_t t = 0; t < nth; t++) {
      const size_t a = start + count * t / nth, b = start + count * (t + 1) / nth;
      // counted per future that exists: the job owns the decrement (a guard object, so a throwing run_range gives it back too);
      // if creating the future itself throws, the guard dies with the unrun closure and the count is restored all the same
      struct Slot {
        bool live = true;
        Slot() { inflight().fetch_add(1, std::memory_order_relaxed); }
        Slot(Slot&& o) noexcept : live(o.live) { o.live = false; }
        Slot(const Slot&) = delete;
        ~Slot() {
          if (live) inflight().fetch_sub(1, std::memory_order_relaxed);
        }
      };
      auto slot = std::make_shared<Slot>();
      futs.emplace_back([dc, a, b, slot]() mutable {
        std::shared_ptr<Slot> mine = std::move(slot);
        return dc.run_range(a, b);
      });
    }
    return futs;
  }
};
struct SliceEntry {
  int curve;
  uint32_t mont;
  const uint8_t* host;  // address of element 0 (identity only; dereferenced solely through a live caller slice)
  size_t n;
  std::shared_ptr<std::vector<uint64_t>> ph;  // ph[i] = hash(point i), all n points
  std::shared_ptr<BaseSet> bs;
  bool tables = false;   // bs carries window tables (built once the array has proved long-lived: third use)
  uint32_t uses = 0;
  uint64_t tick = 0, probes = 0;
  size_t cursor = 0;     // rolling window position (array coordinates; cache_verify = 1)
  // Cheap part, under the cache lock: do the caller's bytes for points [off, off + m) still match?
  bool matches(const uint8_t* slice, size_t off, size_t m, DeepCheck* deep) {
    deep->count = 0;
    if (m == 0) return true;
    const uint64_t* h = ph->data() + off;
    auto ok = [&](size_t i) { return point_hash(slice + 64 * i) == h[i]; };
    if (m <= kFullVerifyBelow) {  // <= 128 KiB: every point, ~10 us
      for (size_t i = 0; i < m; i++)
        if (!ok(i)) return false;
      return true;
    }
    if (!ok(0) || !ok(m - 1)) return false;
    uint64_t x = 0x2545f4914f6cdd1dull * (++probes);
    for (int j = 0; j < 8; j++) {
      x ^= x >> 12, x ^= x << 25, x ^= x >> 27;
      if (!ok((size_t)((x * 0x2545f4914f6cdd1dull) >> 33) % m)) return false;
    }
    deep->ph = ph;
    deep->slice = slice;
    deep->off = off;
    if (G.cache_verify.load(std::memory_order_relaxed) == 0) {  // the whole slice, beside the MSM
      deep->start = 0;
      deep->count = m;
      return true;
    }
    // the rolling window: continues where the last call stopped, clipped to this call's range
    size_t w = std::max(kSyncWindow, m / kWindowFraction);
    if (w > m) w = m;
    size_t st = cursor >= off && cursor < off + m ? cursor - off : 0;
    if (st + w > m) w = m - st;
    cursor = off + st + w >= off + m ? off : off + st + w;
    deep->start = st;
    deep->count = w;
    return true;
  }
};
struct SliceCache {
  std::mutex mu;         // entries, budget
  std::mutex upload_mu;  // one miss at a time: two rayon workers committing to the same key upload it once
  std::list<SliceEntry> entries;
  size_t bytes = 0, max_bytes = 0, min_n = 128, max_entries = 32;
  uint32_t table_after_uses = 2;  // window tables are built when an array is used for the (this + 1)-th time (0: at upload)
  uint64_t clock = 0;
};
static SliceCache& SC = *new SliceCache;  // never destroyed (see G)

static void cache_init_defaults() {
  std::lock_guard<std::mutex> ck(SC.mu);
  if (SC.max_bytes == 0) {
    size_t free_b = 0, total_b = 0;
    HIPCHK(hipMemGetInfo(&free_b, &total_b));
    SC.max_bytes = total_b / 4;
  }
  if (const char* t = getenv("NMX_CACHE_BYTES")) SC.max_bytes = (size_t)atoll(t);
  if (const char* t = getenv("NMX_CACHE_MIN_N")) SC.min_n = (size_t)atoll(t);
  if (const char* t = getenv("NMX_CACHE_TABLE_AFTER")) SC.table_after_uses = (uint32_t)atoi(t);
}
static void cache_publish_gauges() {  // SC.mu held
  g_stats[NMX_STAT_CACHE_ENTRIES].store(SC.entries.size(), std::memory_order_relaxed);
  g_stats[NMX_STAT_CACHE_BYTES].store(SC.bytes, std::memory_order_relaxed);
}
static void cache_erase(std::list<SliceEntry>::iterator it) {  // SC.mu held; in-flight users keep the BaseSet alive
  SC.bytes -= it->bs->bytes();
  SC.entries.erase(it);
}
static void cache_evict_lru_locked() {  // SC.mu held
  auto lru = SC.entries.begin();
  for (auto it = SC.entries.begin(); it != SC.entries.end(); ++it)
    if (it->tick < lru->tick) lru = it;
  cache_erase(lru);
  stat_add(NMX_STAT_CACHE_EVICTIONS);
}
struct SliceKey {
  BaseRef bs;  // null: not cacheable, upload for this call only
  size_t offset = 0;
  DeepCheck deep;       // hit on a long array: the rolling-window check still to run (with_slice)
  bool want_tables = false;  // hit on an entry without tables that has now proved long-lived
};
// hit: the resident key and the offset of `bases` inside it
static bool cache_find(int curve, uint32_t mont, const uint8_t* bases, size_t n, SliceKey* out, bool* grow) {
  std::lock_guard<std::mutex> lk(SC.mu);
  *grow = false;
  for (auto it = SC.entries.begin(); it != SC.entries.end(); ++it) {
    SliceEntry& e = *it;
    if (e.curve != curve || e.mont != mont) continue;
    if (bases < e.host || bases >= e.host + 64 * e.n || ((size_t)(bases - e.host) & 63)) continue;
    const size_t off = (size_t)(bases - e.host) / 64;
    DeepCheck dc;
    if (n > e.n - off) {  // reaches past the resident part
      if (off == 0) {     // a longer prefix of a known array: re-register at the new length
        *grow = e.matches(bases, 0, e.n, &dc);
        cache_erase(it);
        cache_publish_gauges();
        return false;
      }
      continue;
    }
    if (!e.matches(bases, off, n, &dc)) {  // same address, different content: the array was freed and reused
      cache_erase(it);
      cache_publish_gauges();
      return false;
    }
    e.tick = ++SC.clock;
    e.uses++;
    out->bs = e.bs;
    out->offset = off;
    out->deep = dc;
    out->want_tables = !e.tables && e.uses > SC.table_after_uses;
    return true;
  }
  return false;
}
static bool is_oom(const Fail& f) { return f.code == NMX_E_HIP && f.msg.find("out of memory") != std::string::npos; }
// key[0, n) with window tables from a resident copy without them (device to device; shard by shard)
static std::shared_ptr<BaseSet> key_with_tables(const CurveOps& o, const BaseSet& src) {
  auto dst = std::make_shared<BaseSet>(src.curve, src.n);
  if (src.parts.empty()) {
    CtxLease L(src.dev);
    dst->dev = src.dev;
    o.upload(*L.c, *dst, src.d, NMX_BASES_DEVICE | NMX_BASES_INTERNAL | NMX_BASES_PRECOMPUTE, nullptr);
    (void)hipSetDevice(G.device);
    add_prefix_tables(*dst);
    return dst;
  }
  dst->parts.resize(src.parts.size());
  dst->part_begin = src.part_begin;
  for (size_t i = 0; i < src.parts.size(); i++) {
    dst->parts[i] = std::make_shared<BaseSet>(src.curve, src.parts[i]->n);
    dst->parts[i]->dev = src.parts[i]->dev;
  }
  run_on_parts(src.parts.size(), true, [&](size_t i) {
    CtxLease L(src.parts[i]->dev);
    o.upload(*L.c, *dst->parts[i], src.parts[i]->d, NMX_BASES_DEVICE | NMX_BASES_INTERNAL | NMX_BASES_PRECOMPUTE, nullptr);
  });
  add_prefix_tables(*dst->parts[0]);
  return dst;
}
// An entry that has proved long-lived gets its window tables (built from the resident copy: no host traffic).  A failure --
// tables that do not fit the budget or the HBM left -- leaves the plain entry in place: the MSM runs without tables.
static void cache_add_tables(const CurveOps& o, int curve, uint32_t mont, const uint8_t* host0, SliceKey* k) {
  std::lock_guard<std::mutex> up(SC.upload_mu);
  BaseRef plain;
  {
    std::lock_guard<std::mutex> lk(SC.mu);
    for (auto& e : SC.entries)
      if (e.curve == curve && e.mont == mont && e.host == host0 && e.bs == k->bs) {
        if (e.tables) return;  // another thread built them meanwhile; this call runs on the copy it already holds
        plain = e.bs;
      }
  }
  if (!plain) return;
  size_t budget;
  {
    std::lock_guard<std::mutex> lk(SC.mu);
    budget = SC.max_bytes;
  }
  std::shared_ptr<BaseSet> full;
  try {
    if (o.table_bytes(plain->n) > budget) throw Fail{NMX_E_HIP, "window tables beyond the cache budget: out of memory"};
    full = key_with_tables(o, *plain);
  } catch (const Fail& f) {
    if (!is_oom(f)) throw;
    stat_add(NMX_STAT_TABLE_FALLBACKS);
    std::lock_guard<std::mutex> lk(SC.mu);
    for (auto& e : SC.entries)
      if (e.bs == plain) e.tables = true;  // do not try again on every call
    return;
  }
  std::lock_guard<std::mutex> lk(SC.mu);
  for (auto& e : SC.entries)
    if (e.bs == plain) {
      SC.bytes += full->bytes() - plain->bytes();
      e.bs = full;
      e.tables = true;
      k->bs = full;
    }
  while (SC.entries.size() > 1 && SC.bytes > SC.max_bytes) cache_evict_lru_locked();
  cache_publish_gauges();
}
static SliceKey slice_key(Ctx& c, const CurveOps& o, int curve, const void* bases, size_t n, uint32_t flags) {
  SliceKey k;
  if ((flags & (NMX_BASES_NOCACHE | NMX_BASES_DEVICE)) || n == 0) return k;
  const uint32_t mont = (flags & NMX_BASES_MONT) ? 1u : 0u;
  const uint8_t* b = (const uint8_t*)bases;
  bool grow = false;
  if (cache_find(curve, mont, b, n, &k, &grow)) {
    stat_add(NMX_STAT_CACHE_HITS);
    if (k.want_tables) cache_add_tables(o, curve, mont, b - 64 * k.offset, &k);
    return k;
  }
  if (n < SC.min_n) return k;
  std::lock_guard<std::mutex> up(SC.upload_mu);
  bool grow2 = false;
  if (cache_find(curve, mont, b, n, &k, &grow2)) {  // another thread uploaded it while this one waited
    stat_add(NMX_STAT_CACHE_HITS);
    return k;
  }
  size_t budget;
  {
    std::lock_guard<std::mutex> lk(SC.mu);
    budget = SC.max_bytes;
  }
  if (n * 64 > budget) return k;  // does not fit the cache at all: one-shot upload
  SliceEntry e;
  e.curve = curve;
  e.mont = mont;
  e.host = b;
  e.n = n;
  e.ph = std::make_shared<std::vector<uint64_t>>(n);
  hash_points(b, 0, n, e.ph->data());
  // First sight: the key only (upload + conversion).  Tables cost W - 1 more copies of it and ~13 ms per 2^20 points;
  // arrays seen once or twice (IPA's folded keys, src/provider/pedersen.rs:484-497: a fresh n/2-point key per round,
  // two MSMs each) never pay for them.  A re-registration because a longer prefix arrived keeps its standing.
  const bool regrow = grow || grow2;
  const uint32_t up_flags = (flags & NMX_BASES_MONT) | (regrow && SC.table_after_uses <= 2 ? NMX_BASES_PRECOMPUTE : 0u) |
                            (SC.table_after_uses == 0 ? NMX_BASES_PRECOMPUTE : 0u);
  auto upload = [&] {
    return build_key(c, curve, n, true, true, [&](Ctx& cx, BaseSet& part, size_t begin) {
      o.upload(cx, part, (const char*)bases + 64 * begin, up_flags, nullptr);
    });
  };
  for (int attempt = 0;; attempt++) {
    try {
      e.bs = upload();
      break;
    } catch (const Fail& f) {
      // out of HBM: give back what the cache holds, oldest first, and try again; with nothing left to evict the call
      // still runs, on a one-shot upload without tables (temp_key)
      if (!is_oom(f)) throw;
      std::lock_guard<std::mutex> lk(SC.mu);
      if (SC.entries.empty() || attempt >= 64) return k;
      cache_evict_lru_locked();
      cache_publish_gauges();
    }
  }
  e.tables = (up_flags & NMX_BASES_PRECOMPUTE) != 0;
  e.uses = 1;
  stat_add(NMX_STAT_CACHE_UPLOADS);
  if (regrow) stat_add(NMX_STAT_CACHE_REGROWS);
  stat_add(NMX_STAT_BASE_BYTES_H2D, n * 64);
  k.bs = e.bs;
  k.offset = 0;
  std::lock_guard<std::mutex> lk(SC.mu);
  e.tick = ++SC.clock;
  SC.bytes += e.bs->bytes();
  SC.entries.push_back(std::move(e));
  while (SC.entries.size() > 1 && (SC.bytes > SC.max_bytes || SC.entries.size() > SC.max_entries)) cache_evict_lru_locked();
  cache_publish_gauges();
  return k;
}
// One-shot upload for a slice-form call that bypasses the cache (short arrays, NMX_BASES_NOCACHE): plain path, no
// tables; freed when the call returns.
static std::shared_ptr<BaseSet> temp_key(Ctx& c, const CurveOps& o, int curve, const void* bases, size_t n,
                                         uint32_t flags) {
  stat_add(NMX_STAT_UNCACHED_CALLS);
  if (!(flags & NMX_BASES_DEVICE)) stat_add(NMX_STAT_BASE_BYTES_H2D, n * 64);
  return build_key(c, curve, n, false, false, [&](Ctx& cx, BaseSet& part, size_t) {
    o.upload(cx, part, bases, flags & ~(uint32_t)NMX_BASES_PRECOMPUTE, nullptr);
  });
}
// the resident (or one-shot) key behind a slice-form call
static SliceKey resolve_slice(Ctx& c, const CurveOps& o, int curve, const void* bases, size_t n, uint32_t flags) {
  SliceKey k = slice_key(c, o, curve, bases, n, flags);
  if (!k.bs) {
    k.bs = temp_key(c, o, curve, bases, n, flags);
    k.offset = 0;
  }
  return k;
}
// A slice-form call: run(key) computes into the caller's TEMPORARIES; on a long cached array the rolling-window check
// of the caller's bytes runs on a second host thread meanwhile.  A mismatch (the array was edited in place) drops the
// entry and repeats the call on a fresh upload -- the stale result is never handed out.  commit() publishes.
template <class Run, class Commit>
static void with_slice(Ctx& c, const CurveOps& o, int curve, const void* bases, size_t n, uint32_t flags, Run&& run,
                       Commit&& commit) {
  for (int attempt = 0;; attempt++) {
    const SliceKey k = resolve_slice(c, o, curve, bases, n, flags);
    bool ok = true;
    std::vector<PoolFuture<bool>> futs;
    if (k.deep.needed()) {
      if (k.deep.count > kSyncWindow) futs = k.deep.run_async();
      else ok = k.deep.run();
    }
    if (ok) run(k);  // (an exception unwinds through the futures' destructors, which wait for the check)
    for (auto& f : futs) ok = f.get() && ok;
    if (ok) {
      commit();
      return;
    }
    stat_add(NMX_STAT_CACHE_STALE);
    {
      std::lock_guard<std::mutex> lk(SC.mu);
      const uint8_t* b = (const uint8_t*)bases;
      for (auto it = SC.entries.begin(); it != SC.entries.end();) {
        auto cur = it++;
        if (b >= cur->host && b < cur->host + 64 * cur->n) cache_erase(cur);
      }
      cache_publish_gauges();
    }
    require(attempt < 2, NMX_E_ARG, "the base array keeps changing while the call runs");
  }
}


// ---------------------------------------------------------------------------------------------------
// Shard-resident vectors (nmx_svec_*): a field vector laid out like the key it will be committed against -- element i on
// the device that holds key point i -- so that W, E, T are BORN on the shard that commits them (the reference chunks
// coefficients and bases together, /root/reference/src/provider/msm.rs:564-574) and the NIFS fold / cross-term kernels
// (src/r1cs/mod.rs:1044-1107,614-620) run shard by shard, each on its own GPU.
// ---------------------------------------------------------------------------------------------------
struct SVec {
  size_t n_key = 0, n = 0;
  uint32_t k = 1;  // devices of the layout
  struct Piece {
    int dev;
    size_t begin, cnt;
    void* d;
  };
  std::vector<Piece> pieces;  // nmx_shard_plan(n_key, k, 0, n) order
  SVec() = default;
  SVec(const SVec&) = delete;
  SVec& operator=(const SVec&) = delete;
  ~SVec() {
    int prev = -1;
    (void)hipGetDevice(&prev);
    for (Piece& p : pieces)
      if (p.d) {
        (void)hipSetDevice(hip_device_of(p.dev));
        (void)hipFree(p.d);
      }
    if (prev >= 0) (void)hipSetDevice(prev);
  }
  bool same_layout(const SVec& o) const {
    if (n != o.n || pieces.size() != o.pieces.size()) return false;
    for (size_t i = 0; i < pieces.size(); i++)
      if (pieces[i].dev != o.pieces[i].dev || pieces[i].begin != o.pieces[i].begin || pieces[i].cnt != o.pieces[i].cnt) return false;
    return true;
  }
};
static std::unordered_map<uint64_t, std::shared_ptr<SVec>>& svecs() {
  static auto& m = *new std::unordered_map<uint64_t, std::shared_ptr<SVec>>;
  return m;
}
static std::shared_ptr<SVec> svec_lookup(uint64_t h) {
  std::lock_guard<std::mutex> lk(G.mu);
  auto it = svecs().find(h);
  if (it == svecs().end()) throw Fail{NMX_E_HANDLE, "unknown sharded-vector handle"};
  return it->second;
}
// the number of devices a key of n_key points is (or would be) cut over: build_key's rule
static uint32_t shard_count_for(size_t n_key) {
  const uint32_t k = G.ndev_active.load(std::memory_order_relaxed);
  return (k <= 1 || n_key < G.shard_min_n.load(std::memory_order_relaxed) || n_key < k) ? 1u : k;
}
// the pieces of a sharded-scalar call must be the pieces of the key range it addresses
static void check_svec_against_key(const SVec& v, const BaseSet& bs, size_t n) {
  require(n <= v.n, NMX_E_ARG, "sharded vector shorter than the call");
  if (bs.parts.empty()) {
    require(v.pieces.size() <= 1 && (v.pieces.empty() || v.pieces[0].dev == bs.dev), NMX_E_ARG,
            "sharded vector laid out for a multi-device key, the key is on one device");
    return;
  }
  const std::vector<PartJob> jobs = parts_of(bs, 0, n);
  require(jobs.size() <= v.pieces.size(), NMX_E_ARG, "sharded vector layout does not match the key's shards");
  for (size_t i = 0; i < jobs.size(); i++)
    require(jobs[i].part->dev == v.pieces[i].dev && jobs[i].goff == v.pieces[i].begin && jobs[i].cnt <= v.pieces[i].cnt, NMX_E_ARG,
            "sharded vector layout does not match the key's shards (allocate it with the key's length as n_key)");
}

}  // namespace nmx

#include "keyfile.hpp"

using namespace nmx;

extern "C" {

int nmx_init(int device) {
  return guarded([&] {
    {
      std::lock_guard<std::mutex> lk(G.mu);
      if (!G.inited) G.device = device;
    }
    ensure_init();
  });
}

static void drain_pending_commits();  // nmx_commit_begin's tickets (below): shutdown waits for the commitments in flight
int nmx_shutdown(void) {
  return guarded([&] {
    drain_pending_commits();
    std::lock_guard<std::mutex> lk(G.mu);
    if (!G.inited) return;
    (void)hipSetDevice(G.device);
    G.bases.clear();  // the shared_ptr destructors free the HBM
    G.sparse.clear();
    svecs().clear();
    RC.destroy();
    {
      std::lock_guard<std::mutex> ck(SC.mu);
      SC.entries.clear();
      SC.bytes = 0;
      cache_publish_gauges();
    }
    for (Ctx* c : G.all_ctx) {
      (void)hipSetDevice(hip_device_of(c->dev));
      if (c->arena) (void)hipFree(c->arena);
      if (c->aux) (void)hipFree(c->aux);
      if (c->pinned) (void)hipHostFree(c->pinned);
      if (c->mail) (void)hipHostFree(c->mail);
      if (c->have_ev)
        for (int i = 0; i < kMaxMarks; i++) (void)hipEventDestroy(c->ev[i]);
      if (c->async_ev) (void)hipEventDestroy(c->async_ev);
      if (c->side_ev) (void)hipEventDestroy(c->side_ev);
      if (c->chal) (void)hipFree(c->chal);
      for (hipStream_t sd : c->side)
        if (sd) (void)hipStreamDestroy(sd);
      if (c->stream) (void)hipStreamDestroy(c->stream);
      delete c;
    }
    G.all_ctx.clear();
    g_ctx_epoch.fetch_add(1, std::memory_order_acq_rel);
    t_async_ctx = nullptr;
    G.free_ctx.clear();
    G.hip_dev.clear();
    G.ndev_active.store(1);
    G.inited = false;
  });
}

int nmx_sync(void) {
  return guarded([&] {
    if (Ctx* p = async_pending(); p && p->async_ev) HIPCHK(hipEventSynchronize(p->async_ev));
    t_async_ctx = nullptr;
  });
}

int nmx_device_count(void) {
  int cnt = 0;
  if (hipGetDeviceCount(&cnt) != hipSuccess) return 0;
  return cnt;
}

int nmx_init_devices(int count, uint32_t flags) {
  return guarded([&] {
    ensure_init();
    int phys = 0;
    HIPCHK(hipGetDeviceCount(&phys));
    require(count >= 0 && (size_t)count <= Global::kMaxDevices, NMX_E_ARG, "device count out of range");
    if (count == 0) count = phys;
    const bool over = (flags & NMX_DEVICES_OVERSUBSCRIBE) != 0;
    if (count > phys && !over)
      throw Fail{NMX_E_NO_DEVICE, "nmx_init_devices: " + std::to_string(count) + " devices requested, " +
                                      std::to_string(phys) + " visible"};
    std::lock_guard<std::mutex> lk(G.mu);
    // logical device i -> HIP device: 0 is the primary chosen by nmx_init, the others count up from it (mod the visible
    // devices when oversubscribed); existing entries never change, so keys registered earlier stay valid
    for (size_t i = G.hip_dev.size(); i < (size_t)count; i++) G.hip_dev.push_back((G.device + (int)i) % phys);
    for (int i = 0; i < count; i++)
      require(over || G.hip_dev[(size_t)i] == (G.device + i) % phys, NMX_E_ARG, "device map changed between calls");
    if (!over) {  // distinct GPUs only: the map must not wrap
      for (int i = 1; i < count; i++) require(G.hip_dev[(size_t)i] != G.device, NMX_E_NO_DEVICE, "not enough devices after the primary");
    }
    // peer access for the xGMI pulls of HBM-resident scalars (key_msm); best effort: hipMemcpyPeerAsync works without it
    for (int i = 1; i < count; i++) {
      if (G.hip_dev[(size_t)i] == G.device) continue;
      if (hipSetDevice(G.hip_dev[(size_t)i]) == hipSuccess) (void)hipDeviceEnablePeerAccess(G.device, 0);
      (void)hipGetLastError();
    }
    (void)hipSetDevice(G.device);
    G.ndev_active.store((uint32_t)count);
    RC.destroy();  // the communicator follows the device set: rebuilt by the next sharded call
  });
}
int nmx_devices_in_use(void) { return (int)G.ndev_active.load(); }
int nmx_shard_plan(size_t n_key, int k, size_t offset, size_t n, size_t* out_triples, int cap) {
  if (k < 1 || offset > n_key || n > n_key - offset) return NMX_E_ARG;
  int cnt = 0;
  for (int i = 0; i < k && n; i++) {
    const PartRange r = shard_range(n_key, (uint32_t)i, (uint32_t)k);
    const size_t b = r.begin, e = r.begin + r.n;
    const size_t lo = offset > b ? offset : b, hi = offset + n < e ? offset + n : e;
    if (lo >= hi) continue;
    if (out_triples && cnt < cap) {
      out_triples[3 * cnt] = (size_t)i;
      out_triples[3 * cnt + 1] = lo - b;
      out_triples[3 * cnt + 2] = hi - lo;
    }
    cnt++;
  }
  return cnt;
}

int nmx_bases_shard_plan(uint64_t handle, size_t offset, size_t n, size_t* out_triples, int cap, size_t* n_key) {
  int cnt = 0;
  const int rc = guarded([&] {
    auto bs = lookup(handle);
    if (n_key) *n_key = bs->n;
    require(slice_ok(*bs, offset, n), NMX_E_HANDLE, "offset + n beyond the registered key");
    auto put = [&](size_t dev, size_t poff, size_t c) {
      if (out_triples && cnt < cap) {
        out_triples[3 * cnt] = dev;
        out_triples[3 * cnt + 1] = poff;
        out_triples[3 * cnt + 2] = c;
      }
      cnt++;
    };
    if (bs->parts.empty()) {
      if (n) put((size_t)bs->dev, offset, n);
      return;
    }
    for (const PartJob& j : parts_of(*bs, offset, n)) put((size_t)j.part->dev, j.poff, j.cnt);
  });
  return rc ? rc : cnt;
}

const char* nmx_last_error(void) { return t_err.c_str(); }
const char* nmx_version(void) { return "nova-mi355x 0.1.0 (gfx950)"; }

int nmx_bases_register(int curve, const void* bases, size_t n, uint32_t flags, uint64_t* handle) {
  return guarded([&] {
    require(handle && (bases || n == 0), NMX_E_ARG, "null argument");
    const CurveOps& o = ops(curve);
    CtxLease L;
    // (a device-resident source array lives on one device: such keys stay whole)
    *handle = publish(build_key(*L.c, curve, n, !(flags & NMX_BASES_DEVICE), true, [&](Ctx& cx, BaseSet& part, size_t begin) {
      o.upload(cx, part, (const char*)bases + 64 * begin, flags, nullptr);
    }));
  });
}

int nmx_bases_register_ptau(int curve, const char* path, size_t num_g1, size_t num_g2, uint32_t flags, uint64_t* handle) {
  return guarded([&] {
    require(path && handle, NMX_E_ARG, "null argument");
    require(num_g1 < (1ull << 31), NMX_E_TOO_LARGE, "key too large");
    const CurveOps& o = ops(curve);
    FileCloser fc{fopen(path, "rb")};
    if (!fc.f) throw Fail{NMX_E_IO, std::string("IoError: cannot open ") + path};
    // read_ptau (ptau.rs:399-436): meta data, header checks, then the first num_g1 points of section 2
    const PtauMeta meta = ptau_read_meta(fc.f);
    seek_to(fc.f, meta.pos_header);
    ptau_read_header(fc.f, o.base_modulus_words, num_g1, num_g2);
    seek_to(fc.f, meta.pos_tau_g1);
    CtxLease L;
    const uint32_t fl = (flags & NMX_BASES_PRECOMPUTE) | NMX_BASES_MONT | NMX_BASES_VALIDATE;
    // shards are read in file order, one after the other (the file position advances through them)
    *handle = publish(build_key(*L.c, curve, num_g1, true, false, [&](Ctx& cx, BaseSet& part, size_t) {
      const BaseFill fill = file_fill(fc.f, part.n);
      o.upload(cx, part, nullptr, fl, &fill);
    }));
  });
}

int nmx_bases_register_keyfile(int curve, const char* path, size_t n, uint32_t flags, uint64_t* handle,
                               uint8_t* h_xy64) {
  return guarded([&] {
    require(path && handle && h_xy64, NMX_E_ARG, "null argument");
    require(n < (1ull << 31), NMX_E_TOO_LARGE, "key too large");
    const CurveOps& o = ops(curve);
    FileCloser fc{fopen(path, "rb")};
    if (!fc.f) throw Fail{NMX_E_IO, std::string("IoError: cannot open ") + path};
    char head[12];
    read_exact(fc.f, head, 12, "head");
    require(memcmp(head, "PEDERSEN_KEY", 12) == 0, NMX_E_FORMAT, "InvalidHead");  // pedersen.rs:324-330
    // points[0] = h, points[1..] = ck (pedersen.rs:332-339)
    uint8_t h_raw[64], h_canon[64];
    read_exact(fc.f, h_raw, 64, "h");
    require(o.check_point_host(h_raw, NMX_BASES_MONT, h_canon), NMX_E_POINT,
            "PointNotOnCurve: h is not canonical or not on the curve");
    CtxLease L;
    const uint32_t fl = (flags & NMX_BASES_PRECOMPUTE) | NMX_BASES_MONT | NMX_BASES_VALIDATE;
    auto bs = build_key(*L.c, curve, n, true, false, [&](Ctx& cx, BaseSet& part, size_t) {
      const BaseFill fill = file_fill(fc.f, part.n);
      o.upload(cx, part, nullptr, fl, &fill);
    });
    memcpy(h_xy64, h_canon, 64);
    *handle = publish(std::move(bs));
  });
}

int nmx_bases_unregister(uint64_t handle) {
  return guarded([&] {
    ensure_init();
    std::shared_ptr<BaseSet> bs;  // dropped outside the lock; the HBM is freed when the last in-flight call lets go
    {
      std::lock_guard<std::mutex> lk(G.mu);
      auto it = G.bases.find(handle);
      if (it == G.bases.end()) throw Fail{NMX_E_HANDLE, "unknown base handle"};
      bs = std::move(it->second);
      G.bases.erase(it);
    }
  });
}

int nmx_bases_read(uint64_t handle, size_t offset, size_t n, void* out_xy64) {
  return guarded([&] {
    require(out_xy64 || n == 0, NMX_E_ARG, "null argument");
    auto bs = lookup(handle);
    require(slice_ok(*bs, offset, n), NMX_E_HANDLE, "offset + n beyond the registered key");
    auto copy_back = [&](const BaseSet& part, size_t poff, size_t cnt, size_t goff) {
      CtxLease L(part.dev);
      HIPCHK(hipMemcpyAsync((char*)out_xy64 + goff * 64, (const char*)part.d + poff * 64, cnt * 64, hipMemcpyDeviceToHost,
                            L.c->stream));
      HIPCHK(hipStreamSynchronize(L.c->stream));
    };
    if (bs->parts.empty()) {
      if (n) copy_back(*bs, offset, n, 0);
    } else {
      for (const PartJob& j : parts_of(*bs, offset, n)) copy_back(*j.part, j.poff, j.cnt, j.goff);
      (void)hipSetDevice(G.device);
    }
    ops(bs->curve).internal_to_canonical((uint8_t*)out_xy64, 2 * n);
  });
}

int nmx_bases_generate(int curve, uint64_t k0, size_t n, uint32_t flags, uint64_t* handle) {
  return guarded([&] {
    require(handle != nullptr, NMX_E_ARG, "null argument");
    require(n < (1ull << 31) && k0 < (1ull << 62), NMX_E_ARG, "k0 / n out of range");
    const CurveOps& o = ops(curve);
    CtxLease L;
    *handle = publish(build_key(*L.c, curve, n, true, true, [&](Ctx& cx, BaseSet& part, size_t begin) {
      o.generate(cx, part, k0 + begin, flags);
    }));
  });
}

int nmx_msm_handle(uint64_t handle, size_t offset, const void* scalars, size_t n, uint32_t flags,
                   uint8_t* out, uint8_t* out_is_inf) {
  return guarded([&] {
    require(out && (scalars || n == 0), NMX_E_ARG, "null argument");
    auto bs = lookup(handle);
    require(slice_ok(*bs, offset, n), NMX_E_HANDLE, "offset + n beyond the registered key");
    CtxLease L;
    stat_add(NMX_STAT_MSM_CALLS);
    key_msm(*L.c, *bs, offset, n, field_call(scalars, flags), flags, out, out_is_inf);
  });
}

int nmx_msm(int curve, const void* scalars, const void* bases, size_t n, uint32_t flags, uint8_t* out,
            uint8_t* out_is_inf) {
  return guarded([&] {
    require(out && ((scalars && bases) || n == 0), NMX_E_ARG, "null argument");
    const CurveOps& o = ops(curve);
    CtxLease L;
    stat_add(NMX_STAT_MSM_CALLS);
    uint8_t res[128], rinf = 0;
    with_slice(*L.c, o, curve, bases, n, flags,
               [&](const SliceKey& k) { key_msm(*L.c, *k.bs, k.offset, n, field_call(scalars, flags), flags, res, &rinf); },
               [&] {
                 memcpy(out, res, (flags & NMX_OUT_PARTIAL) ? 128 : 64);
                 if (out_is_inf) *out_is_inf = rinf;
               });
  });
}

int nmx_msm_u64_handle(uint64_t handle, size_t offset, const uint64_t* scalars, size_t n,
                       uint32_t max_num_bits, uint32_t flags, uint8_t* out, uint8_t* out_is_inf) {
  return guarded([&] {
    require(out && (scalars || n == 0), NMX_E_ARG, "null argument");
    auto bs = lookup(handle);
    require(slice_ok(*bs, offset, n), NMX_E_HANDLE, "offset + n beyond the registered key");
    CtxLease L;
    const bool sharded = (flags & NMX_SCALARS_SHARDED) != 0;
    require(!sharded || max_num_bits != NMX_BITS_AUTO, NMX_E_ARG, "sharded small scalars need an explicit max_num_bits");
    bool dev = (flags & NMX_SCALARS_DEVICE) != 0;
    uint32_t bits = resolve_u64_bits(*L.c, scalars, n, dev, max_num_bits);
    MsmCall mc{scalars, dev || sharded, false, bits, true};
    mc.scalars_sharded = sharded;
    stat_add(NMX_STAT_MSM_CALLS);
    key_msm(*L.c, *bs, offset, n, mc, flags, out, out_is_inf);
  });
}

int nmx_msm_u64(int curve, const uint64_t* scalars, const void* bases, size_t n, uint32_t max_num_bits,
                uint32_t flags, uint8_t* out, uint8_t* out_is_inf) {
  return guarded([&] {
    require(out && ((scalars && bases) || n == 0), NMX_E_ARG, "null argument");
    const CurveOps& o = ops(curve);
    CtxLease L;
    bool dev = (flags & NMX_SCALARS_DEVICE) != 0;
    uint32_t bits = resolve_u64_bits(*L.c, scalars, n, dev, max_num_bits);
    MsmCall mc{scalars, dev, false, bits, true};
    stat_add(NMX_STAT_MSM_CALLS);
    uint8_t res[128], rinf = 0;
    with_slice(*L.c, o, curve, bases, n, flags,
               [&](const SliceKey& k) { key_msm(*L.c, *k.bs, k.offset, n, mc, flags, res, &rinf); },
               [&] {
                 memcpy(out, res, (flags & NMX_OUT_PARTIAL) ? 128 : 64);
                 if (out_is_inf) *out_is_inf = rinf;
               });
  });
}

int nmx_msm_sparse_handle(uint64_t handle, const uint64_t* indices, const void* scalars, size_t k, uint32_t flags,
                          uint8_t* out, uint8_t* out_is_inf) {
  return guarded([&] {
    require(out && (indices || k == 0), NMX_E_ARG, "null argument");
    require(!(flags & NMX_SCALARS_DEVICE) || scalars, NMX_E_ARG, "device flag without scalars");
    auto bs = lookup(handle);
    std::vector<uint32_t> idx(k ? k : 1);
    for (size_t i = 0; i < k; i++) {
      require(indices[i] < bs->n, NMX_E_HANDLE, "index beyond the registered key");  // ck.ck[i] would panic
      idx[i] = (uint32_t)indices[i];
    }
    CtxLease L;
    stat_add(NMX_STAT_MSM_CALLS);
    key_msm_sparse(*L.c, *bs, idx.data(), scalars, k, flags, out, out_is_inf);
  });
}

// The reference's default is `scalars.par_iter().map(msm)` (traits.rs:82-90).  Here the shortest vectors (as many as the
// key's window width leaves key bits for: 32 on a 2^14..2^19-point key, 16 at 2^20..2^21) are FUSED into one pipeline run over the key's
// tables, one bucket set per vector -- an MSM of a few thousand pairs costs ~0.3 ms of dependent point additions
// whatever its length, and the fused run pays that once (profiles/r02_msm_2p20/batch_fused.txt).  The remaining
// (longest) vectors and the fused run are jobs taken longest-first by up to kBatchLanes host threads, each leasing its
// own context (stream + workspace): independent runs overlap on the GPU -- the latency-bound fold / reduction passes
// of one under the accumulate kernel of another.
static constexpr size_t kBatchLanes = 4;
// small_bits: 0 = field scalars; otherwise every vector holds u64 scalars of at most that many bits (NMX_BITS_AUTO: per vector,
// from the data) -- batch_vartime_multiscalar_mul_small, traits.rs:109-117: vector by vector over the resident key, on up to
// kBatchLanes streams (the fused run carries field scalars only).
static void batch_impl(const BaseSet& bs, size_t base_off, size_t n_bases, const void* const* vecs, const size_t* lens,
                       size_t k, uint32_t flags, uint8_t* out, uint8_t* out_is_inf, Ctx& c, uint32_t small_bits = 0,
                       bool use_prefix = true) {
  require((vecs && lens && out) || k == 0, NMX_E_ARG, "null argument");
  require(!(flags & NMX_OUT_PARTIAL), NMX_E_ARG, "NMX_OUT_PARTIAL is not supported for batches");
  const CurveOps& o = ops(bs.curve);
  for (size_t j = 0; j < k; j++) {
    require(lens[j] <= n_bases, NMX_E_ARG, "vector longer than the base array");  // traits.rs:88 slices bases[..len]
    require(vecs[j] || lens[j] == 0, NMX_E_ARG, "null scalar vector");
  }
  // shard-resident scalars over a single-device key (or the prefix of shard 0): every vector has ONE piece, on the key's device
  std::vector<const void*> unwrapped;
  if ((flags & NMX_SCALARS_SHARDED) && bs.parts.empty() && !small_bits) {
    unwrapped.resize(k ? k : 1);
    for (size_t j = 0; j < k; j++) {
      unwrapped[j] = lens[j] ? ((const void* const*)vecs[j])[0] : nullptr;
      if (lens[j]) check_shard_piece(unwrapped[j], lens[j] * 32, bs.dev);
    }
    vecs = unwrapped.data();
    flags = (flags & ~(uint32_t)NMX_SCALARS_SHARDED) | NMX_SCALARS_DEVICE;
  }
  // Vectors that stay inside the key's narrow prefix tables run there, fused (BaseSet::prefix): on a wide-table key nothing
  // fuses (two bucket sets of 2^19 at most) and every short vector pays a 2^19-bucket reduction -- and over a sharded key every
  // vector is a sharded MSM of its own.  The long vectors keep the path below (use_prefix = false on the same key).
  if (use_prefix && !small_bits && k >= 1) {
    const BaseSet* pre = bs.parts.empty() ? bs.prefix.get() : bs.parts[0]->prefix.get();
    if (pre && base_off < pre->n) {
      std::vector<size_t> in, rest;
      for (size_t j = 0; j < k; j++) (base_off + lens[j] <= pre->n ? in : rest).push_back(j);
      if (!in.empty()) {
        auto sub = [&](Ctx& cx, const BaseSet& key, size_t nb, const std::vector<size_t>& idx, std::vector<uint8_t>& o_xy, std::vector<uint8_t>& o_inf) {
          std::vector<const void*> v(idx.size());
          std::vector<size_t> l(idx.size());
          for (size_t q = 0; q < idx.size(); q++) v[q] = vecs[idx[q]], l[q] = lens[idx[q]];
          o_xy.assign(64 * (idx.size() ? idx.size() : 1), 0);
          o_inf.assign(idx.size() ? idx.size() : 1, 0);
          if (!idx.empty()) batch_impl(key, base_off, nb, v.data(), l.data(), idx.size(), flags, o_xy.data(), o_inf.data(), cx, 0, &key != &bs);
        };
        std::vector<uint8_t> xa, ia, xb, ib;
        // the two halves are independent: the vectors of the prefix (which may descend further) on a helper thread with a context
        // of its own, the long ones here
        run_on_parts(rest.empty() ? 1 : 2, true, [&](size_t t) {
          if (t == 0) {
            if (rest.empty()) sub(c, *pre, pre->n - base_off, in, xa, ia);
            else sub(c, bs, n_bases, rest, xb, ib);
          } else {
            CtxLease L2(pre->dev);
            sub(*L2.c, *pre, pre->n - base_off, in, xa, ia);
          }
        });
        for (size_t q = 0; q < in.size(); q++) memcpy(out + 64 * in[q], xa.data() + 64 * q, 64);
        for (size_t q = 0; q < rest.size(); q++) memcpy(out + 64 * rest[q], xb.data() + 64 * q, 64);
        if (out_is_inf) {
          for (size_t q = 0; q < in.size(); q++) out_is_inf[in[q]] = ia[q];
          for (size_t q = 0; q < rest.size(); q++) out_is_inf[rest[q]] = ib[q];
        }
        return;
      }
    }
  }
  auto call_for = [&](Ctx& cx, size_t j) -> MsmCall {
    if (!small_bits) return field_call(vecs[j], flags);
    const bool dev = (flags & NMX_SCALARS_DEVICE) != 0;
    const uint32_t bits = resolve_u64_bits(cx, (const uint64_t*)vecs[j], lens[j], dev, small_bits);
    return MsmCall{vecs[j], dev, false, bits, true};
  };
  if (!bs.parts.empty()) {  // multi-device key: every vector is a sharded MSM of its own, all devices busy with each
    std::vector<uint8_t> tmp(64 * (k ? k : 1)), tinf(k ? k : 1);
    for (size_t j = 0; j < k; j++) {
      stat_add(NMX_STAT_MSM_CALLS);
      key_msm(c, bs, base_off, lens[j], call_for(c, j), flags, tmp.data() + 64 * j, tinf.data() + j);
    }
    memcpy(out, tmp.data(), 64 * k);
    if (out_is_inf) memcpy(out_is_inf, tinf.data(), k);
    return;
  }
  std::vector<size_t> order(k);
  for (size_t j = 0; j < k; j++) order[j] = j;
  std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return lens[a] > lens[b]; });
  // jobs: runs of consecutive vectors in `order` -- one vector = an MSM of its own, several = one fused run.  Groups are
  // cut from the short end, as many vectors each as the key takes; jobs are taken longest-first.
  struct Job {
    size_t first, count;
    uint64_t weight;
  };
  std::vector<Job> jobs;
  {
    const size_t limit = (k >= 2 && !small_bits) ? o.batch_limit(bs) : 0;
    size_t hi = k;
    while (hi > 0) {
      size_t lo = hi - 1;
      uint64_t sum = lens[order[lo]];
      while (limit >= 2 && lo > 0 && hi - lo < limit && (sum + lens[order[lo - 1]]) * bs.pre_W < 0xfff00000ull)
        sum += lens[order[--lo]];  // (entry indices of a run are 32 bits)
      jobs.push_back(Job{lo, hi - lo, sum});
      hi = lo;
    }
    std::stable_sort(jobs.begin(), jobs.end(), [](const Job& x, const Job& y) { return x.weight > y.weight; });
  }
  const size_t njobs = jobs.size();
  // results are staged so that a failure in any vector leaves `out` untouched
  std::vector<uint8_t> tmp(64 * (k ? k : 1)), tinf(k ? k : 1);
  std::atomic<size_t> next{0};
  std::mutex err_mu;
  bool failed = false;
  Fail first_fail{0, ""};
  auto record = [&](const Fail& f) {
    std::lock_guard<std::mutex> lk(err_mu);
    if (!failed) first_fail = f;
    failed = true;
  };
  // nothing may escape a worker: an exception leaving a std::thread body is std::terminate in the host process
  auto guarded_worker = [&](Ctx* ctx) {
    try {
      for (;;) {
        size_t i = next.fetch_add(1);
        if (i >= njobs) return;
        const Job& job = jobs[i];
        if (job.count > 1) {
          const size_t m = job.count;
          std::vector<BatchItem> items(m);
          for (size_t q = 0; q < m; q++) items[q] = BatchItem{vecs[order[job.first + q]], lens[order[job.first + q]]};
          std::vector<uint8_t> r(64 * m), rinf(m);
          stat_add(NMX_STAT_MSM_CALLS, m);
          stat_add(NMX_STAT_FUSED_RUNS);
          o.msm_key_batch(*ctx, bs, base_off, items.data(), m, field_call(nullptr, flags), flags, r.data(), rinf.data());
          for (size_t q = 0; q < m; q++) {
            memcpy(tmp.data() + 64 * order[job.first + q], r.data() + 64 * q, 64);
            tinf[order[job.first + q]] = rinf[q];
          }
          continue;
        }
        const size_t j = order[job.first];
        stat_add(NMX_STAT_MSM_CALLS);
        o.msm_key(*ctx, bs, base_off, lens[j], call_for(*ctx, j), flags, tmp.data() + 64 * j, tinf.data() + j);
      }
    } catch (const Fail& f) {
      record(f);
    } catch (const std::exception& e) {
      record(Fail{NMX_E_HIP, e.what()});
    } catch (...) {
      record(Fail{NMX_E_HIP, "unknown exception in a batch worker"});
    }
  };
  const size_t lanes = njobs < kBatchLanes ? njobs : kBatchLanes;
  // lane 0 on the calling thread with its context, the others on pool workers with a context of their own; a lane that cannot
  // be started (no thread) is simply missing: the others drain the job list
  run_on_parts(lanes, true, [&](size_t t) {
    if (t == 0) {
      guarded_worker(&c);
    } else {
      CtxLease L;
      guarded_worker(L.c);
    }
  });
  if (failed) throw first_fail;
  memcpy(out, tmp.data(), 64 * k);
  if (out_is_inf) memcpy(out_is_inf, tinf.data(), k);
}

int nmx_msm_batch_handle(uint64_t handle, const void* const* scalar_vecs, const size_t* lens, size_t k,
                         uint32_t flags, uint8_t* out, uint8_t* out_is_inf) {
  return guarded([&] {
    auto bs = lookup(handle);
    CtxLease L;
    batch_impl(*bs, 0, bs->n, scalar_vecs, lens, k, flags, out, out_is_inf, *L.c);
  });
}

int nmx_msm_batch(int curve, const void* const* scalar_vecs, const size_t* lens, size_t k,
                  const void* bases, size_t n_bases, uint32_t flags, uint8_t* out, uint8_t* out_is_inf) {
  return guarded([&] {
    require(bases || n_bases == 0, NMX_E_ARG, "null argument");
    const CurveOps& o = ops(curve);
    CtxLease L;
    if (n_bases == 0) {
      BaseSet empty(curve, 0);
      batch_impl(empty, 0, 0, scalar_vecs, lens, k, flags, out, out_is_inf, *L.c);
      return;
    }
    require((out && lens) || k == 0, NMX_E_ARG, "null argument");
    std::vector<uint8_t> res(64 * (k ? k : 1)), rinf(k ? k : 1);
    with_slice(*L.c, o, curve, bases, n_bases, flags,
               [&](const SliceKey& key) {
                 batch_impl(*key.bs, key.offset, n_bases, scalar_vecs, lens, k, flags, res.data(), rinf.data(), *L.c);
               },
               [&] {
                 memcpy(out, res.data(), 64 * k);
                 if (out_is_inf) memcpy(out_is_inf, rinf.data(), k);
               });
  });
}

int nmx_msm_u64_batch_handle(uint64_t handle, const uint64_t* const* scalar_vecs, const size_t* lens, size_t k, uint32_t max_num_bits,
                             uint32_t flags, uint8_t* out, uint8_t* out_is_inf) {
  return guarded([&] {
    require(max_num_bits == NMX_BITS_AUTO || max_num_bits <= 64, NMX_E_ARG, "max_num_bits must be <= 64");
    require(!(flags & NMX_SCALARS_SHARDED), NMX_E_ARG, "batches of small scalars take plain pointers");
    auto bs = lookup(handle);
    CtxLease L;
    if (max_num_bits == 0) {  // msm.rs:489: every result is the identity
      require((out && lens) || k == 0, NMX_E_ARG, "null argument");
      for (size_t j = 0; j < k; j++) require(lens[j] <= bs->n, NMX_E_ARG, "vector longer than the base array");
      memset(out, 0, 64 * k);
      if (out_is_inf) memset(out_is_inf, 1, k);
      return;
    }
    batch_impl(*bs, 0, bs->n, (const void* const*)scalar_vecs, lens, k, flags, out, out_is_inf, *L.c, max_num_bits);
  });
}

int nmx_msm_u64_batch(int curve, const uint64_t* const* scalar_vecs, const size_t* lens, size_t k, const void* bases, size_t n_bases,
                      uint32_t max_num_bits, uint32_t flags, uint8_t* out, uint8_t* out_is_inf) {
  return guarded([&] {
    require(bases || n_bases == 0, NMX_E_ARG, "null argument");
    require(max_num_bits == NMX_BITS_AUTO || max_num_bits <= 64, NMX_E_ARG, "max_num_bits must be <= 64");
    require(!(flags & NMX_SCALARS_SHARDED), NMX_E_ARG, "batches of small scalars take plain pointers");
    require((out && lens) || k == 0, NMX_E_ARG, "null argument");
    const CurveOps& o = ops(curve);
    CtxLease L;
    if (n_bases == 0 || max_num_bits == 0) {
      for (size_t j = 0; j < k; j++) require(lens[j] <= n_bases, NMX_E_ARG, "vector longer than the base array");
      if (k) memset(out, 0, 64 * k);
      if (out_is_inf) memset(out_is_inf, 1, k);
      return;
    }
    std::vector<uint8_t> res(64 * (k ? k : 1)), rinf(k ? k : 1);
    with_slice(*L.c, o, curve, bases, n_bases, flags,
               [&](const SliceKey& key) {
                 batch_impl(*key.bs, key.offset, n_bases, (const void* const*)scalar_vecs, lens, k, flags, res.data(), rinf.data(), *L.c,
                            max_num_bits);
               },
               [&] {
                 memcpy(out, res.data(), 64 * k);
                 if (out_is_inf) memcpy(out_is_inf, rinf.data(), k);
               });
  });
}

static void commit_impl(CtxLease& L, const BaseSet& bs, MsmCall mc, size_t n, const void* h_xy64, const void* r, uint32_t flags,
                        uint8_t* out, uint8_t* out_is_inf) {
  require(n <= bs.n, NMX_E_HANDLE, "ck shorter than v");  // assert!(ck.ck.len() >= v.len()), pedersen.rs:264
  stat_add(NMX_STAT_MSM_CALLS);
  const CurveOps& o = ops(bs.curve);
  if (bs.parts.empty()) {
    if (mc.scalars_sharded) {  // one piece, on the key's device
      mc.scalars = n ? ((const void* const*)mc.scalars)[0] : nullptr;
      require(mc.scalars || n == 0, NMX_E_ARG, "null shard pointer");
      if (n) check_shard_piece(mc.scalars, n * (mc.u64_mode ? 8u : 32u), bs.dev);  // as key_msm / batch_impl: NMX_E_ARG, not a GPU fault
      mc.scalars_device = true;
      mc.scalars_sharded = false;
    }
    o.commit(*L.c, prefix_or_key(bs, 0, n), n, mc, h_xy64, r, flags, out, out_is_inf);
  } else {  // sharded key: the blinding term is one more partial, computed on the host under the device MSMs
    uint8_t two[256];
    std::array<uint8_t, 64> hb;
    std::array<uint8_t, 32> rb;
    memcpy(hb.data(), h_xy64, 64);
    memcpy(rb.data(), r, 32);
    PoolFuture<std::array<uint8_t, 128>> hr([&o, hb, rb, flags] {
      std::array<uint8_t, 128> t;
      o.blind_term(hb.data(), rb.data(), flags, t.data());
      return t;
    });
    key_msm(*L.c, bs, 0, n, mc, flags | NMX_OUT_PARTIAL, two, nullptr);
    const auto t = hr.get();  // (an out-of-range r surfaces here, NMX_E_SCALAR_RANGE, before anything is written)
    memcpy(two + 128, t.data(), 128);
    o.point_sum(two, 2, flags, out, out_is_inf);
  }
}
int nmx_commit(uint64_t ck_handle, const void* v, size_t n, const void* h_xy64, const void* r,
               uint32_t flags, uint8_t* out, uint8_t* out_is_inf) {
  return guarded([&] {
    require(out && (v || n == 0) && h_xy64 && r, NMX_E_ARG, "null argument");
    auto bs = lookup(ck_handle);
    CtxLease L;
    commit_impl(L, *bs, field_call(v, flags), n, h_xy64, r, flags, out, out_is_inf);
  });
}

// nmx_commit_begin / nmx_commit_finish: a commitment that runs BESIDE the caller's next calls.  The two MSMs of a folding step do
// not depend on each other -- commit_T reads W2 and X, never comm_W (src/r1cs/mod.rs:590-622); the RO absorbs comm_W
// (nifs.rs:53) but is squeezed only behind comm_T (:60-63) -- and each of them ends in a latency-bound tail (fold, reduce tree)
// that leaves the chip idle: side by side, one's tail hides under the other's accumulation.  The commitment runs on a pool
// worker with a context leased HERE (so its stream is ordered behind this thread's asynchronous calls, like any other call).
struct PendingCommit {
  struct Res {
    std::array<uint8_t, 128> out{};
    uint8_t inf = 0;
  };
  PoolFuture<Res> fut;
  uint32_t flags = 0;
};
static std::mutex g_pend_mu;
static std::unordered_map<uint64_t, std::shared_ptr<PendingCommit>>& pending_commits() {
  static auto* m = new std::unordered_map<uint64_t, std::shared_ptr<PendingCommit>>();
  return *m;
}
static uint64_t g_next_ticket = 1;
static void drain_pending_commits() {
  std::unordered_map<uint64_t, std::shared_ptr<PendingCommit>> gone;
  {
    std::lock_guard<std::mutex> lk(g_pend_mu);
    gone.swap(pending_commits());
  }
  gone.clear();  // ~PoolFuture waits; results and errors of unfinished tickets are dropped
}
int nmx_commit_begin(uint64_t ck_handle, const void* v, size_t n, const void* h_xy64, const void* r, uint32_t flags, uint64_t* ticket) {
  return guarded([&] {
    require(ticket && (v || n == 0) && h_xy64 && r, NMX_E_ARG, "null argument");
    BaseRef bs = lookup(ck_handle);
    require(n <= bs->n, NMX_E_HANDLE, "ck shorter than v");
    std::array<uint8_t, 64> hb;
    std::array<uint8_t, 32> rb;
    memcpy(hb.data(), h_xy64, 64);
    memcpy(rb.data(), r, 32);
    const MsmCall mc = field_call(v, flags);
    auto lease = std::make_shared<CtxLease>();  // on the calling thread: behind its stream-ordered calls
    // The worker must see the caller's pending NMX_ASYNC mark as well (thread-local): a sharded key leases one context per shard
    // INSIDE the job (key_msm -> run_on_parts reads async_pending() of the thread it runs on), and those leases have to wait for
    // the caller's asynchronous producer just as the lease above does (capi.hip:1914).
    Ctx* const pend = async_pending();
    const uint64_t pend_epoch = t_async_epoch;
    auto pc = std::make_shared<PendingCommit>();
    pc->flags = flags;
    pc->fut = PoolFuture<PendingCommit::Res>([bs, mc, n, hb, rb, flags, lease, pend, pend_epoch]() mutable {
      std::shared_ptr<CtxLease> L = std::move(lease);  // handed back when the job ends, whatever happens
      struct Restore {  // the worker's own mark comes back when the job ends (as run_on_parts' helper threads do)
        Ctx* v;
        uint64_t e;
        ~Restore() { t_async_ctx = v, t_async_epoch = e; }
      } restore{t_async_ctx, t_async_epoch};
      t_async_ctx = pend, t_async_epoch = pend_epoch;
      HIPCHK(hipSetDevice(hip_device_of(L->c->dev)));  // (the device is per host thread)
      PendingCommit::Res res;
      commit_impl(*L, *bs, mc, n, hb.data(), rb.data(), flags, res.out.data(), &res.inf);
      return res;
    });
    std::lock_guard<std::mutex> lk(g_pend_mu);
    const uint64_t t = g_next_ticket++;
    pending_commits()[t] = std::move(pc);
    *ticket = t;
  });
}
int nmx_commit_finish(uint64_t ticket, uint8_t* out, uint8_t* out_is_inf) {
  return guarded([&] {
    require(out != nullptr, NMX_E_ARG, "null argument");  // (before the ticket is touched: it stays valid)
    std::shared_ptr<PendingCommit> pc;
    {
      std::lock_guard<std::mutex> lk(g_pend_mu);
      auto it = pending_commits().find(ticket);
      if (it == pending_commits().end()) throw Fail{NMX_E_HANDLE, "unknown commitment ticket"};
      pc = std::move(it->second);
      pending_commits().erase(it);
    }
    const PendingCommit::Res res = pc->fut.get();  // rethrows what the commitment threw: the error is reported HERE
    memcpy(out, res.out.data(), (pc->flags & NMX_OUT_PARTIAL) ? 128 : 64);
    if (out_is_inf) *out_is_inf = res.inf;
  });
}

// ---- shard-resident vectors ---------------------------------------------------------------------------
int nmx_svec_alloc(size_t n_key, size_t n, uint64_t* handle) {
  return guarded([&] {
    require(handle != nullptr, NMX_E_ARG, "null argument");
    require(n <= n_key && n_key < (1ull << 31), NMX_E_ARG, "n must be <= n_key < 2^31");
    ensure_init();
    auto v = std::make_shared<SVec>();
    v->n_key = n_key;
    v->n = n;
    v->k = shard_count_for(n_key);
    for (uint32_t i = 0; i < v->k; i++) {
      const PartRange r = shard_range(n_key, i, v->k);
      const size_t lo = r.begin, hi = r.begin + r.n < n ? r.begin + r.n : n;
      if (lo < hi) v->pieces.push_back(SVec::Piece{(int)i, lo, hi - lo, nullptr});
    }
    for (SVec::Piece& p : v->pieces) {  // (a failure frees what was allocated: ~SVec)
      HIPCHK(hipSetDevice(hip_device_of(p.dev)));
      HIPCHK(hipMalloc(&p.d, p.cnt * 32));
    }
    (void)hipSetDevice(G.device);
    std::lock_guard<std::mutex> lk(G.mu);
    const uint64_t h = G.next_handle++;
    svecs()[h] = std::move(v);
    *handle = h;
  });
}
int nmx_svec_free(uint64_t handle) {
  return guarded([&] {
    std::shared_ptr<SVec> v;  // freed outside the lock, when the last in-flight call lets go
    {
      std::lock_guard<std::mutex> lk(G.mu);
      auto it = svecs().find(handle);
      if (it == svecs().end()) throw Fail{NMX_E_HANDLE, "unknown sharded-vector handle"};
      v = std::move(it->second);
      svecs().erase(it);
    }
  });
}
// raw 32-byte elements in / out: no conversion (the field kernels work on canonical or Montgomery data alike)
static void svec_copy(const SVec& v, void* host, bool to_device) {
  run_on_parts(v.pieces.size(), true, [&](size_t i) {
    const SVec::Piece& p = v.pieces[i];
    CtxLease L(p.dev);
    char* h = (char*)host + 32 * p.begin;
    if (to_device) HIPCHK(hipMemcpyAsync(p.d, h, p.cnt * 32, hipMemcpyHostToDevice, L.c->stream));
    else HIPCHK(hipMemcpyAsync(h, p.d, p.cnt * 32, hipMemcpyDeviceToHost, L.c->stream));
    HIPCHK(hipStreamSynchronize(L.c->stream));
  });
}
int nmx_svec_write(uint64_t handle, const void* host) {
  return guarded([&] {
    auto v = svec_lookup(handle);
    require(host || v->n == 0, NMX_E_ARG, "null argument");
    svec_copy(*v, const_cast<void*>(host), true);
  });
}
int nmx_svec_read(uint64_t handle, void* host) {
  return guarded([&] {
    auto v = svec_lookup(handle);
    require(host || v->n == 0, NMX_E_ARG, "null argument");
    svec_copy(*v, host, false);
  });
}
int nmx_svec_parts(uint64_t handle, void** dev_ptrs, size_t* counts, int* devices, int cap) {
  int cnt = 0;
  const int rc = guarded([&] {
    auto v = svec_lookup(handle);
    cnt = (int)v->pieces.size();
    for (int i = 0; i < cnt && i < cap; i++) {
      if (dev_ptrs) dev_ptrs[i] = v->pieces[(size_t)i].d;
      if (counts) counts[i] = v->pieces[(size_t)i].cnt;
      if (devices) devices[i] = hip_device_of(v->pieces[(size_t)i].dev);
    }
  });
  return rc == NMX_OK ? cnt : rc;
}
int nmx_svec_map(int field, int op, const uint64_t* in, int n_in, const void* challenge, uint32_t flags, uint64_t out) {
  return guarded([&] {
    static const int need_in[] = {2, 3, 4, 5, 2};  // AXPY, AXPY2, CROSS_TERM, CROSS_TERM2, VEC_ADD
    require(op >= 0 && op <= NMX_OP_VEC_ADD && in && n_in == need_in[op], NMX_E_ARG, "bad operation / operand count");
    require(challenge || op == NMX_OP_VEC_ADD, NMX_E_ARG, "null challenge");
    require(!(flags & ~(uint32_t)NMX_SCALARS_MONT), NMX_E_ARG, "only NMX_SCALARS_MONT applies");
    std::vector<std::shared_ptr<SVec>> v((size_t)n_in);
    for (int j = 0; j < n_in; j++) v[(size_t)j] = svec_lookup(in[j]);
    auto o = svec_lookup(out);
    for (auto& x : v) require(x->same_layout(*o), NMX_E_ARG, "operands with different shard layouts");
    const uint32_t fl = flags | NMX_SCALARS_DEVICE;
    run_on_parts(o->pieces.size(), true, [&](size_t i) {
      const size_t m = o->pieces[i].cnt;
      CtxLease L(o->pieces[i].dev);
      auto d = [&](int j) { return v[(size_t)j]->pieces[i].d; };
      void* od = o->pieces[i].d;
      switch (op) {
        case NMX_OP_AXPY: fv_axpy(*L.c, field, d(0), d(1), challenge, m, fl, od); break;
        case NMX_OP_AXPY2: fv_axpy2(*L.c, field, d(0), d(1), d(2), challenge, m, fl, od); break;
        case NMX_OP_CROSS_TERM: fv_cross_term(*L.c, field, d(0), d(1), d(2), d(3), challenge, m, fl, od); break;
        case NMX_OP_CROSS_TERM2: fv_cross_term2(*L.c, field, d(0), d(1), d(2), d(3), d(4), challenge, m, fl, od); break;
        default: fv_vec_add(*L.c, field, d(0), d(1), m, fl, od); break;
      }
    });
  });
}
// the pointer array a sharded-scalar call takes, from a sharded vector (checked against the key's shards)
static std::vector<const void*> svec_pointers(const SVec& v, const BaseSet& bs, size_t n) {
  check_svec_against_key(v, bs, n);
  std::vector<const void*> p;
  for (const SVec::Piece& q : v.pieces) p.push_back(q.d);
  if (p.empty()) p.push_back(nullptr);
  return p;
}
int nmx_msm_svec(uint64_t key_handle, uint64_t svec, size_t n, uint32_t flags, uint8_t* out, uint8_t* out_is_inf) {
  return guarded([&] {
    require(out != nullptr, NMX_E_ARG, "null argument");
    auto bs = lookup(key_handle);
    auto v = svec_lookup(svec);
    require(slice_ok(*bs, 0, n), NMX_E_HANDLE, "n beyond the registered key");
    const std::vector<const void*> ptrs = svec_pointers(*v, *bs, n);
    const uint32_t fl = (flags & ~(uint32_t)NMX_SCALARS_DEVICE) | NMX_SCALARS_SHARDED;
    CtxLease L;
    stat_add(NMX_STAT_MSM_CALLS);
    key_msm(*L.c, *bs, 0, n, field_call(ptrs.data(), fl), fl, out, out_is_inf);
  });
}
int nmx_commit_svec(uint64_t ck_handle, uint64_t svec, size_t n, const void* h_xy64, const void* r, uint32_t flags,
                    uint8_t* out, uint8_t* out_is_inf) {
  return guarded([&] {
    require(out && h_xy64 && r, NMX_E_ARG, "null argument");
    auto bs = lookup(ck_handle);
    auto v = svec_lookup(svec);
    require(n <= bs->n, NMX_E_HANDLE, "ck shorter than v");
    const std::vector<const void*> ptrs = svec_pointers(*v, *bs, n);
    const uint32_t fl = (flags & ~(uint32_t)NMX_SCALARS_DEVICE) | NMX_SCALARS_SHARDED;
    CtxLease L;
    commit_impl(L, *bs, field_call(ptrs.data(), fl), n, h_xy64, r, fl, out, out_is_inf);
  });
}
int nmx_profile_last_sharded(float* ms, int* dev, int* branch, int cap, float* combine_ms, int* rccl_ranks) {
  const int n = (int)t_shards.size();
  for (int i = 0; i < n && i < cap; i++) {
    const ShardRec& r = t_shards[(size_t)i];
    if (ms)
      for (int q = 0; q < kMaxMarks; q++) ms[i * kMaxMarks + q] = q < r.nst ? r.ms[q] : 0.0f;
    if (dev) dev[i] = r.dev;
    if (branch) branch[i] = r.branch;
  }
  if (combine_ms) *combine_ms = t_combine_ms;
  if (rccl_ranks) *rccl_ranks = t_rccl_ranks;
  return n;
}

int nmx_point_sum(int curve, const uint8_t* partials128, size_t count, uint8_t* out, uint8_t* out_is_inf) {
  return guarded([&] {
    require(out && (partials128 || count == 0), NMX_E_ARG, "null argument");
    ops(curve).point_sum(partials128, count, 0, out, out_is_inf);
  });
}

// ---- field-vector kernels -------------------------------------------------------
int nmx_field_axpy(int field, const void* a, const void* b, const void* r, size_t n, uint32_t flags, void* out) {
  return guarded([&] {
    require((a && b && out) || n == 0, NMX_E_ARG, "null argument");
    require(r != nullptr, NMX_E_ARG, "null argument");
    if (n == 0) return;
    CtxLease L;
    fv_axpy(*L.c, field, a, b, r, n, flags, out);
  });
}
int nmx_field_axpy2(int field, const void* a, const void* b, const void* c, const void* r, size_t n, uint32_t flags,
                    void* out) {
  return guarded([&] {
    require((a && b && c && out) || n == 0, NMX_E_ARG, "null argument");
    require(r != nullptr, NMX_E_ARG, "null argument");
    if (n == 0) return;
    CtxLease L;
    fv_axpy2(*L.c, field, a, b, c, r, n, flags, out);
  });
}
int nmx_field_cross_term(int field, const void* az, const void* bz, const void* cz, const void* e, const void* u,
                         size_t n, uint32_t flags, void* out) {
  return guarded([&] {
    require((az && bz && cz && e && out) || n == 0, NMX_E_ARG, "null argument");
    require(u != nullptr, NMX_E_ARG, "null argument");
    if (n == 0) return;
    CtxLease L;
    fv_cross_term(*L.c, field, az, bz, cz, e, u, n, flags, out);
  });
}
int nmx_field_cross_term2(int field, const void* az, const void* bz, const void* cz, const void* e1, const void* e2,
                          const void* u, size_t n, uint32_t flags, void* out) {
  return guarded([&] {
    require((az && bz && cz && e1 && e2 && out) || n == 0, NMX_E_ARG, "null argument");
    require(u != nullptr, NMX_E_ARG, "null argument");
    if (n == 0) return;
    CtxLease L;
    fv_cross_term2(*L.c, field, az, bz, cz, e1, e2, u, n, flags, out);
  });
}
int nmx_field_vec_add(int field, const void* a, const void* b, size_t n, uint32_t flags, void* out) {
  return guarded([&] {
    require((a && b && out) || n == 0, NMX_E_ARG, "null argument");
    if (n == 0) return;
    CtxLease L;
    fv_vec_add(*L.c, field, a, b, n, flags, out);
  });
}
int nmx_mle_bind_top(int field, const void* z, size_t len, const void* r, uint32_t flags, void* out) {
  return guarded([&] {
    require(z && out && r, NMX_E_ARG, "null argument");
    require(len >= 2 && (len & 1) == 0, NMX_E_ARG, "len must be even and >= 2");  // assert!(self.num_vars > 0)
    CtxLease L;
    fv_bind(*L.c, field, z, len, 0, len / 2, 1, r, len / 2, flags, out);
  });
}
int nmx_poly_fold_pairs(int field, const void* p, size_t len, const void* x, uint32_t flags, void* out) {
  return guarded([&] {
    require(p && out && x, NMX_E_ARG, "null argument");
    require(len >= 2 && (len & 1) == 0, NMX_E_ARG, "len must be even and >= 2");
    require(!((flags & NMX_SCALARS_DEVICE) && out == p), NMX_E_ARG, "pairwise fold cannot run in place");
    CtxLease L;
    fv_bind(*L.c, field, p, len, 0, 1, 2, x, len / 2, flags, out);
  });
}

int nmx_poly_fold_chain(int field, const void* p, size_t len, const void* xs, size_t k, uint32_t flags, void* const* outs) {
  return guarded([&] {
    require(p && (xs || k == 0) && (outs || k == 0), NMX_E_ARG, "null argument");
    require(field >= 0 && field < 4, NMX_E_ARG, "bad field id");
    require(len >= 2 && (len & (len - 1)) == 0 && len < (1ull << 32), NMX_E_ARG, "len must be a power of two >= 2");
    require(k >= 1 && (len >> k) >= 1, NMX_E_ARG, "at most log2(len) folds");
    for (size_t i = 0; i < k; i++) require(outs[i] && outs[i] != p, NMX_E_ARG, "null output, or a fold in place");
    if (!(flags & NMX_SCALARS_DEVICE)) {  // host vectors: fold by fold through the single-fold path
      const void* cur = p;
      for (size_t i = 0; i < k; i++) {
        const int rc = nmx_poly_fold_pairs(field, cur, len >> i, (const uint8_t*)xs + 32 * i, flags, outs[i]);
        if (rc) throw Fail{rc, nmx_last_error()};
        cur = outs[i];
      }
      return;
    }
    CtxLease L;
    fv_fold_chain(*L.c, field, p, len, xs, k, flags, outs);
  });
}

int nmx_sumcheck_eq_sums(int field, int mode, const void* A, const void* B, const void* C, size_t len, const void* eqL,
                         size_t n_eqL, const void* eqR, size_t n_eqR, uint32_t shift, uint32_t flags, uint8_t* out64) {
  return guarded([&] {
    require(A && eqR && out64, NMX_E_ARG, "null argument");
    require(mode >= 1 && mode <= 3 && (mode < 2 || B) && (mode < 3 || C), NMX_E_ARG, "missing input polynomial");
    require(len >= 2 && (len & 1) == 0 && len / 2 < (1ull << 31), NMX_E_ARG, "len must be even");
    const size_t h = len / 2;
    if (eqL) {
      require(shift < 32 && n_eqR == ((size_t)1 << shift) && ((h - 1) >> shift) < n_eqL, NMX_E_ARG,
              "eq tables do not cover the index range");  // poly_eq_right.len() == 1 << second_half (sumcheck.rs:1238)
    } else {
      require(n_eqR >= h, NMX_E_ARG, "eq table shorter than the half length");
    }
    CtxLease L;
    fv_eq_sums(*L.c, field, mode, A, B, C, len, eqL, n_eqL, eqR, n_eqR, shift, flags, out64);
  });
}

int nmx_sumcheck_bind_eq_sums(int field, int mode, const void* A, const void* B, const void* C, size_t len, const void* r,
                              const void* eqL, size_t n_eqL, const void* eqR, size_t n_eqR, uint32_t shift,
                              uint32_t flags, void* outA, void* outB, void* outC, uint8_t* out64) {
  return guarded([&] {
    require(A && outA && r && eqR && out64, NMX_E_ARG, "null argument");
    require(mode >= 1 && mode <= 3 && (mode < 2 || (B && outB)) && (mode < 3 || (C && outC)), NMX_E_ARG,
            "missing input / output polynomial");
    require(flags & NMX_SCALARS_DEVICE, NMX_E_ARG, "the fused round works on HBM-resident tables only");
    require(len >= 4 && (len & 3) == 0 && len / 4 < (1ull << 31), NMX_E_ARG, "len must be a multiple of 4");
    const size_t hq = len / 4;  // the next round's half length
    if (eqL) {
      require(shift < 32 && n_eqR == ((size_t)1 << shift) && ((hq - 1) >> shift) < n_eqL, NMX_E_ARG,
              "eq tables do not cover the next round's index range");
    } else {
      require(n_eqR >= hq, NMX_E_ARG, "eq table shorter than the next round's half length");
    }
    CtxLease L;
    fv_bind_eq_sums(*L.c, field, mode, A, B, C, len, r, eqL, n_eqL, eqR, n_eqR, shift, flags, outA, outB, outC, out64);
  });
}

int nmx_sumcheck_plain_sums(int field, int kind, const void* A, const void* B, const void* C, size_t len, uint32_t flags,
                            uint8_t* out96) {
  return guarded([&] {
    require(A && B && out96, NMX_E_ARG, "null argument");
    require(kind >= 1 && kind <= 4 && (kind < 4 || C), NMX_E_ARG, "missing input polynomial");
    require(len >= 2 && (len & 1) == 0 && len / 2 < (1ull << 31), NMX_E_ARG, "len must be even");
    CtxLease L;
    uint8_t three[96] = {0};
    fv_plain_sums(*L.c, field, kind, A, B, C, len, flags, three);
    memcpy(out96, three, 96);
  });
}

int nmx_field_lincomb_powers(int field, const void* const* vecs, const size_t* lens, size_t k, const void* s, size_t n_out,
                             uint32_t flags, void* out) {
  return guarded([&] {
    require(s && (out || n_out == 0) && (k == 0 || (vecs && lens)), NMX_E_ARG, "null argument");
    require(k <= 4096 && n_out < (1ull << 31), NMX_E_TOO_LARGE, "too many / too long vectors");
    for (size_t j = 0; j < k; j++) {
      require(lens[j] <= n_out, NMX_E_ARG, "output shorter than an input polynomial");
      require(vecs[j] || lens[j] == 0, NMX_E_ARG, "null vector");
    }
    if (n_out == 0) return;
    CtxLease L;
    fv_lincomb(*L.c, field, vecs, lens, k, s, n_out, flags, out);
  });
}

// The two sqrt-size eq tables of evaluate_with (multilinear.rs:98-129) and the staging copy of a host polynomial,
// carved from the context's aux arena (the main arena is re-carved by fv_eq_sums)
struct EvalScratch {
  uint32_t *eqL, *eqR;
  void* z;
  size_t s_left, s_right;
  EvalScratch(Ctx& c, int field, const void* r, size_t ell, size_t len, uint32_t flags) {
    s_right = ell / 2;
    s_left = ell - s_right;
    auto pad = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t bl = pad(((size_t)1 << s_left) * 32), br = pad(((size_t)1 << s_right) * 32);
    const size_t bz = (flags & NMX_SCALARS_DEVICE) ? 0 : pad(len * 32);
    aux_reserve(c, bl + br + bz);
    eqL = (uint32_t*)c.aux;
    eqR = (uint32_t*)(c.aux + bl);
    z = bz ? c.aux + bl + br : nullptr;
    fv_eq_evals_pair(c, field, r, (uint32_t)s_left, (uint32_t)s_right, flags, eqL, eqR);
  }
  // sum_id z[id] * eqL[id >> s_right] * eqR[id & (2^s_right - 1)]: the mode-1 sum over "half" = len
  void evaluate(Ctx& c, int field, const void* zp, size_t len, uint32_t flags, uint8_t* out32) {
    const void* dz = zp;
    if (z) {
      HIPCHK(hipMemcpyAsync(z, zp, len * 32, hipMemcpyHostToDevice, c.stream));
      dz = z;
    }
    uint8_t two[64];
    fv_eq_sums(c, field, 1, dz, nullptr, nullptr, 2 * len, eqL, (size_t)1 << s_left, eqR, (size_t)1 << s_right,
               (uint32_t)s_right, flags | NMX_SCALARS_DEVICE, two);
    memcpy(out32, two, 32);
  }
};

int nmx_poly_suffix_horner(int field, const void* f, size_t n, const void* u, uint32_t flags, void* out) {
  return guarded([&] {
    require(f && u && out, NMX_E_ARG, "null argument");
    require(n >= 1 && n < (1ull << 31), NMX_E_ARG, "assert!(!f.is_empty())");
    // never in place, never overlapping: out[i] depends on every f[k >= i] while other waves are still reading them -- and the
    // two-pass kernels the call repeats on after a scan time-out re-read f after the aborted scan has written to out
    {
      const char *fb = (const char*)f, *ob = (const char*)out;
      require(!((flags & NMX_SCALARS_DEVICE) && fb < ob + n * 32 && ob < fb + n * 32), NMX_E_ARG,
              "nmx_poly_suffix_horner cannot run in place (f and out overlap)");
    }
    CtxLease L;
    fv_suffix_horner(*L.c, field, f, n, u, flags, out);
  });
}

int nmx_poly_eval_multi(int field, const void* const* polys, const size_t* lens, size_t k, const void* points, size_t m,
                        uint32_t flags, uint8_t* out) {
  return guarded([&] {
    require((k == 0 || (polys && lens)) && (m == 0 || points) && (out || k * m == 0), NMX_E_ARG, "null argument");
    require(m <= 4 && k <= 4096, NMX_E_TOO_LARGE, "at most 4 points and 4096 polynomials per call");
    for (size_t i = 0; i < k; i++) {
      require(polys[i] || lens[i] == 0, NMX_E_ARG, "null polynomial");
      require(lens[i] < (1ull << 31), NMX_E_TOO_LARGE, "polynomial too long");
    }
    if (k == 0 || m == 0) return;
    CtxLease L;
    fv_eval_multi(*L.c, field, polys, lens, k, points, m, flags, out);
  });
}

int nmx_eq_evals_from_points(int field, const void* r, size_t ell, uint32_t flags, void* out) {
  return guarded([&] {
    require((r || ell == 0) && out, NMX_E_ARG, "null argument");
    require(ell < 31, NMX_E_ARG, "too many variables");
    CtxLease L;
    const size_t n = (size_t)1 << ell;
    if (flags & NMX_SCALARS_DEVICE) {
      fv_eq_evals(*L.c, field, r, (uint32_t)ell, flags, (uint32_t*)out);
      HIPCHK(hipStreamSynchronize(L.c->stream));
    } else {
      aux_reserve(*L.c, n * 32);
      fv_eq_evals(*L.c, field, r, (uint32_t)ell, flags, (uint32_t*)L.c->aux);
      HIPCHK(hipMemcpyAsync(out, L.c->aux, n * 32, hipMemcpyDeviceToHost, L.c->stream));
      HIPCHK(hipStreamSynchronize(L.c->stream));
    }
  });
}

int nmx_mle_evaluate(int field, const void* z, size_t len, const void* r, size_t ell, uint32_t flags, uint8_t* out32) {
  return guarded([&] {
    require(z && (r || ell == 0) && out32, NMX_E_ARG, "null argument");
    require(ell < 31 && len == ((size_t)1 << ell), NMX_E_ARG, "assert_eq!(r.len(), self.get_num_vars())");
    CtxLease L;
    EvalScratch es(*L.c, field, r, ell, len, flags);
    if ((flags & NMX_SCALARS_DEVICE) && ell >= 1) fv_mle_multi_eval(*L.c, field, &z, 1, len, es.eqL, es.eqR, (uint32_t)es.s_right, flags, out32);
    else es.evaluate(*L.c, field, z, len, flags, out32);
  });
}

int nmx_mle_multi_evaluate(int field, const void* const* zs, size_t k, size_t len, const void* r, size_t ell,
                           uint32_t flags, uint8_t* out) {
  return guarded([&] {
    require((zs || k == 0) && (r || ell == 0) && (out || k == 0), NMX_E_ARG, "null argument");
    require(ell < 31 && len == ((size_t)1 << ell), NMX_E_ARG, "assert!(Zs.iter().all(|z| z.len() == n))");
    if (k == 0) return;  // multilinear.rs:133-135
    for (size_t j = 0; j < k; j++) require(zs[j], NMX_E_ARG, "null polynomial");
    CtxLease L;
    // the two sqrt-size eq tables are built once and shared by all k polynomials (multilinear.rs:141-147)
    EvalScratch es(*L.c, field, r, ell, len, flags);
    if ((flags & NMX_SCALARS_DEVICE) && ell >= 1) {  // resident polynomials: all passes enqueued at once, results through the mailbox
      for (size_t j = 0; j < k; j += 16)
        fv_mle_multi_eval(*L.c, field, zs + j, k - j < 16 ? k - j : 16, len, es.eqL, es.eqR, (uint32_t)es.s_right, flags, out + 32 * j);
      return;
    }
    for (size_t j = 0; j < k; j++) es.evaluate(*L.c, field, zs[j], len, flags, out + 32 * j);
  });
}

int nmx_spmv_register(int field, const uint64_t* indptr, const uint64_t* indices, const void* data, size_t rows,
                      size_t cols, uint32_t flags, uint64_t* handle) {
  return guarded([&] {
    require(indptr && handle && field >= 0 && field < 4, NMX_E_ARG, "bad argument");
    const size_t nnz = (size_t)indptr[rows];
    require((indices && data) || nnz == 0, NMX_E_ARG, "null argument");
    require(rows < (1ull << 31) && cols < (1ull << 31) && nnz < (1ull << 31), NMX_E_TOO_LARGE, "matrix too large");
    std::vector<uint32_t> ip(rows + 1), ix(nnz ? nnz : 1);
    for (size_t i = 0; i <= rows; i++) {
      require(indptr[i] <= nnz && (i == 0 || indptr[i] >= indptr[i - 1]), NMX_E_ARG, "indptr not monotone");
      ip[i] = (uint32_t)indptr[i];
    }
    for (size_t k = 0; k < nnz; k++) {
      require(indices[k] < cols, NMX_E_ARG, "column index out of range");
      ix[k] = (uint32_t)indices[k];
    }
    CtxLease L;
    auto sp = std::make_shared<Global::SparseSet>();  // a failure below frees what was allocated (destructor)
    Global::SparseSet& ss = *sp;
    ss.field = field, ss.rows = rows, ss.cols = cols, ss.nnz = nnz;
    HIPCHK(hipMalloc((void**)&ss.indptr, (rows + 1) * 4));
    HIPCHK(hipMalloc((void**)&ss.indices, (nnz ? nnz : 1) * 4));
    HIPCHK(hipMalloc((void**)&ss.data, (nnz ? nnz : 1) * 32));
    HIPCHK(hipMemcpyAsync(ss.indptr, ip.data(), (rows + 1) * 4, hipMemcpyHostToDevice, L.c->stream));
    HIPCHK(hipMemcpyAsync(ss.indices, ix.data(), nnz * 4, hipMemcpyHostToDevice, L.c->stream));
    HIPCHK(hipMemcpyAsync(ss.data, data, nnz * 32, hipMemcpyHostToDevice, L.c->stream));
    fv_spmv_convert(*L.c, field, ss.data, nnz, flags);
    fv_spmv_classify(*L.c, field, ss.data, ss.indices, nnz, cols);  // +-1 / small coefficients: class bits in the index
    HIPCHK(hipStreamSynchronize(L.c->stream));
    std::lock_guard<std::mutex> lk(G.mu);
    uint64_t h = G.next_handle++;
    G.sparse[h] = std::move(sp);
    *handle = h;
  });
}
int nmx_spmv_unregister(uint64_t handle) {
  return guarded([&] {
    std::shared_ptr<Global::SparseSet> sp;  // freed when the last in-flight apply lets go
    {
      std::lock_guard<std::mutex> lk(G.mu);
      auto it = G.sparse.find(handle);
      if (it == G.sparse.end()) throw Fail{NMX_E_HANDLE, "unknown matrix handle"};
      sp = std::move(it->second);
      G.sparse.erase(it);
    }
  });
}
int nmx_spmv_apply(uint64_t handle, const void* z, size_t z_len, uint32_t flags, void* out) {
  return guarded([&] {
    require(z && out, NMX_E_ARG, "null argument");
    std::shared_ptr<Global::SparseSet> sp;
    {
      std::lock_guard<std::mutex> lk(G.mu);
      auto it = G.sparse.find(handle);
      if (it == G.sparse.end()) throw Fail{NMX_E_HANDLE, "unknown matrix handle"};
      sp = it->second;
    }
    const Global::SparseSet& ss = *sp;
    require(z_len == ss.cols, NMX_E_ARG, "invalid shape");  // assert_eq!(self.cols, vector.len(), "invalid shape")
    if (ss.rows == 0) return;
    CtxLease L;
    fv_spmv_apply(*L.c, ss.field, ss.indptr, ss.indices, ss.data, ss.rows, ss.cols, z, flags, out);
  });
}

// M^T in virtual rows, built once per matrix from the resident CSR (Global::SparseSet::Transposed): counting sort of the entries
// by column on the host, columns longer than 32 entries cut into chunks of 16
static std::shared_ptr<Global::SparseSet::Transposed> transposed_of(Ctx& c, Global::SparseSet& ss) {
  std::lock_guard<std::mutex> lk(ss.t_mu);
  if (ss.tr) return ss.tr;
  const size_t rows = ss.rows, cols = ss.cols, nnz = ss.nnz;
  std::vector<uint32_t> ip(rows + 1), ix(nnz ? nnz : 1);
  std::vector<uint8_t> dt((nnz ? nnz : 1) * 32);
  HIPCHK(hipMemcpyAsync(ip.data(), ss.indptr, (rows + 1) * 4, hipMemcpyDeviceToHost, c.stream));
  if (nnz) {
    HIPCHK(hipMemcpyAsync(ix.data(), ss.indices, nnz * 4, hipMemcpyDeviceToHost, c.stream));
    HIPCHK(hipMemcpyAsync(dt.data(), ss.data, nnz * 32, hipMemcpyDeviceToHost, c.stream));
  }
  HIPCHK(hipStreamSynchronize(c.stream));
  const bool tagged_in = cols <= ((size_t)1 << 28), tagged_out = rows <= ((size_t)1 << 28);
  const uint32_t cmask = tagged_in ? (1u << 28) - 1u : 0xffffffffu;
  std::vector<uint32_t> cnt(cols + 1, 0);
  for (size_t k = 0; k < nnz; k++) cnt[(ix[k] & cmask) + 1]++;
  for (size_t j = 0; j < cols; j++) cnt[j + 1] += cnt[j];
  std::vector<uint32_t> tix(nnz ? nnz : 1), pos(cnt.begin(), cnt.end() - 1);
  std::vector<uint8_t> tdt((nnz ? nnz : 1) * 32);
  for (size_t r = 0; r < rows; r++)
    for (uint32_t k = ip[r]; k < ip[r + 1]; k++) {
      const uint32_t col = ix[k] & cmask, cls = tagged_in ? ix[k] >> 28 : 0u, q = pos[col]++;
      tix[q] = (uint32_t)r | (tagged_out ? cls << 28 : 0u);  // the class rides along (same coefficient); none if rows need all 32 bits
      memcpy(tdt.data() + 32 * (size_t)q, dt.data() + 32 * (size_t)k, 32);
    }
  std::vector<uint32_t> vptr{0}, vout, hrow, hstart{0};
  size_t nparts = 0;
  for (size_t j = 0; j < cols; j++) {
    const uint32_t b = cnt[j], e = cnt[j + 1], L = e - b;
    if (L <= 32) {
      vptr.push_back(e);
      vout.push_back((uint32_t)j);
      continue;
    }
    // every lane walks at most 16 entries of a split column -- each step is a gather whose latency nothing hides, 64-entry chunks
    // made the lanes of ONE long column the longest thing in the launch (profiles/r05_spartan: 214 us against 62 for a matrix
    // without such a column); a split column's partials are summed by a block (k_spmv_heavy)
    const uint32_t T = 16;
    for (uint32_t a = b; a < e; a += T) {
      vptr.push_back(a + T < e ? a + T : e);
      vout.push_back(0x80000000u | (uint32_t)nparts++);
    }
    hrow.push_back((uint32_t)j);
    hstart.push_back((uint32_t)nparts);
  }
  require(nparts < (1ull << 31), NMX_E_TOO_LARGE, "matrix too large");
  auto tr = std::make_shared<Global::SparseSet::Transposed>();
  tr->dev = c.dev;
  auto up = [&](uint32_t** d, const void* h, size_t bytes) {
    HIPCHK(hipMalloc((void**)d, bytes ? bytes : 4));
    if (bytes) HIPCHK(hipMemcpyAsync(*d, h, bytes, hipMemcpyHostToDevice, c.stream));
  };
  up(&tr->vptr, vptr.data(), vptr.size() * 4);
  up(&tr->indices, tix.data(), nnz * 4);
  up(&tr->data, tdt.data(), nnz * 32);
  up(&tr->vout, vout.data(), vout.size() * 4);
  up(&tr->hrow, hrow.data(), hrow.size() * 4);
  up(&tr->hstart, hstart.data(), hstart.size() * 4);
  HIPCHK(hipStreamSynchronize(c.stream));  // the host vectors go away
  tr->nvirt = vout.size(), tr->nheavy = hrow.size(), tr->nparts = nparts;
  tr->dev = c.dev;
  ss.tr = tr;
  return tr;
}
int nmx_spmv_apply_transposed(uint64_t handle, const void* x, size_t x_len, uint32_t flags, void* out) {
  return guarded([&] {
    require(x && out, NMX_E_ARG, "null argument");
    std::shared_ptr<Global::SparseSet> sp;
    {
      std::lock_guard<std::mutex> lk(G.mu);
      auto it = G.sparse.find(handle);
      if (it == G.sparse.end()) throw Fail{NMX_E_HANDLE, "unknown matrix handle"};
      sp = it->second;
    }
    Global::SparseSet& ss = *sp;
    require(x_len == ss.rows, NMX_E_ARG, "invalid shape");  // assert_eq!(rx.len(), S.num_cons()), spartan/mod.rs:504
    if (ss.cols == 0) return;
    CtxLease L;
    auto tr = transposed_of(*L.c, ss);
    fv_spmv_apply_transposed(*L.c, ss.field, tr->vptr, tr->indices, tr->data, tr->vout, tr->hrow, tr->hstart, tr->nvirt, tr->nheavy,
                             tr->nparts, ss.rows, ss.cols, x, flags, out);
  });
}

int nmx_spmv_apply_many(const uint64_t* handles, size_t k, int transposed, const void* x, size_t x_len, uint32_t flags, void* const* outs) {
  return guarded([&] {
    require(handles && x && outs && k >= 1 && k <= 8, NMX_E_ARG, "bad argument (1 .. 8 matrices)");
    std::vector<std::shared_ptr<Global::SparseSet>> sp(k);
    {
      std::lock_guard<std::mutex> lk(G.mu);
      for (size_t i = 0; i < k; i++) {
        auto it = G.sparse.find(handles[i]);
        if (it == G.sparse.end()) throw Fail{NMX_E_HANDLE, "unknown matrix handle"};
        sp[i] = it->second;
      }
    }
    for (size_t i = 0; i < k; i++) {
      require(outs[i], NMX_E_ARG, "null output");
      require(sp[i]->field == sp[0]->field, NMX_E_ARG, "matrices over different fields");
      require(x_len == (transposed ? sp[i]->rows : sp[i]->cols), NMX_E_ARG, "invalid shape");
    }
    if (!(flags & NMX_SCALARS_DEVICE)) {  // host operands: one matrix after the other through the single-matrix paths
      for (size_t i = 0; i < k; i++) {
        const int rc = transposed ? nmx_spmv_apply_transposed(handles[i], x, x_len, flags, outs[i]) : nmx_spmv_apply(handles[i], x, x_len, flags, outs[i]);
        if (rc) throw Fail{rc, nmx_last_error()};
      }
      return;
    }
    CtxLease L;
    std::vector<std::shared_ptr<Global::SparseSet::Transposed>> tr(k);
    std::vector<SpmvManyItem> items(k);
    for (size_t i = 0; i < k; i++) {
      Global::SparseSet& ss = *sp[i];
      SpmvManyItem& m = items[i];
      m.rows = ss.rows, m.cols = ss.cols, m.out = outs[i];
      if (transposed) {
        tr[i] = transposed_of(*L.c, ss);
        m.vptr = tr[i]->vptr, m.tix = tr[i]->indices, m.tdata = tr[i]->data, m.vout = tr[i]->vout, m.hrow = tr[i]->hrow, m.hstart = tr[i]->hstart;
        m.nvirt = tr[i]->nvirt, m.nheavy = tr[i]->nheavy, m.nparts = tr[i]->nparts;
      } else {
        m.indptr = ss.indptr, m.indices = ss.indices, m.data = ss.data;
      }
    }
    fv_spmv_many(*L.c, sp[0]->field, items.data(), k, transposed != 0, x, flags);
  });
}

int nmx_field_batch_invert(int field, const void* v, size_t n, uint32_t flags, void* out) {
  return guarded([&] {
    require(field >= 0 && field < 4, NMX_E_ARG, "bad field id");
    require((v && out) || n == 0, NMX_E_ARG, "null argument");
    require(n < (1ull << 31), NMX_E_TOO_LARGE, "vector too long");
    if (n == 0) return;
    if (flags & NMX_SCALARS_DEVICE) {
      const char *a = (const char*)v, *b = (const char*)out;
      require(!(a < b + n * 32 && b < a + n * 32), NMX_E_ARG, "nmx_field_batch_invert cannot run in place (v and out overlap)");
    }
    CtxLease L;
    if (!fv_batch_invert(*L.c, field, v, n, flags, out)) throw Fail{NMX_E_ZERO, "batch_invert: an element is zero (NovaError::InternalError)"};
  });
}

// z = [W, u, X] zero-padded (snark.rs:133, 193-196) and the clones of batch_eval_reduce (spartan/mod.rs:407-410) for vectors that
// live in HBM: copies on the library's stream, so that they are ordered with the kernels that read them
int nmx_field_concat(int field, const void* const* parts, const size_t* lens, uint64_t device_mask, size_t k, size_t n_out, uint32_t flags,
                     void* out) {
  return guarded([&] {
    require(field >= 0 && field < 4, NMX_E_ARG, "bad field id");
    require((parts && lens) || k == 0, NMX_E_ARG, "null argument");
    require(out || n_out == 0, NMX_E_ARG, "null argument");
    require(flags & NMX_SCALARS_DEVICE, NMX_E_ARG, "nmx_field_concat writes an HBM-resident vector (NMX_SCALARS_DEVICE)");
    require(k <= 64, NMX_E_ARG, "at most 64 parts");
    size_t total = 0;
    for (size_t i = 0; i < k; i++) {
      require(parts[i] || lens[i] == 0, NMX_E_ARG, "null part");
      total += lens[i];
    }
    require(total <= n_out, NMX_E_ARG, "parts longer than the output");
    if (n_out == 0) return;
    CtxLease L;
    size_t off = 0;
    bool host_part = false;
    for (size_t i = 0; i < k; i++) {
      if (!lens[i]) continue;
      const bool dev = (device_mask >> i) & 1u;
      host_part |= !dev;
      HIPCHK(hipMemcpyAsync((char*)out + off * 32, parts[i], lens[i] * 32, dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, L.c->stream));
      off += lens[i];
    }
    if (off < n_out) HIPCHK(hipMemsetAsync((char*)out + off * 32, 0, (n_out - off) * 32, L.c->stream));
    if ((flags & NMX_ASYNC) && !host_part) {
      async_mark(*L.c);
      return;
    }
    stream_wait(L.c->stream);
  });
}

// Spartan's sum-check provers as one call each (sumcheck_prove.hpp)
static void check_sc_args(int field, size_t num_rounds, uint32_t flags, nmx_transcript_fn cb) {
  (void)flags;
  require(field >= 0 && field < 4, NMX_E_ARG, "bad field id");
  require(cb != nullptr, NMX_E_ARG, "null transcript callback");
  require(num_rounds < 31, NMX_E_TOO_LARGE, "too many rounds");
}
// Host tables (no NMX_SCALARS_DEVICE): the reference's `&mut MultilinearPolynomial` are host Vecs, consumed by the prover; they
// are uploaded for the call (their host copies are left as they were) -- the form a shim starts with before it keeps vectors in HBM.
struct ScStaged {
  std::vector<void*> dev;
  ~ScStaged() {
    for (void* p : dev)
      if (p) (void)hipFree(p);
  }
  void* up(Ctx& c, const void* host, size_t elems) {
    void* d = nullptr;
    HIPCHK(hipMalloc(&d, elems * 32 ? elems * 32 : 32));
    dev.push_back(d);
    HIPCHK(hipMemcpyAsync(d, host, elems * 32, hipMemcpyHostToDevice, c.stream));
    return d;
  }
};
int nmx_sumcheck_prove_cubic_with_three_inputs(int field, const void* claim, const void* taus, size_t num_rounds, void* A, void* B, void* C,
                                               uint32_t flags, nmx_transcript_fn transcript, void* ctx, uint8_t* out_polys, uint8_t* out_r,
                                               uint8_t* out_claims) {
  return guarded([&] {
    check_sc_args(field, num_rounds, flags, transcript);
    require(claim && (taus || num_rounds == 0) && A && B && C, NMX_E_ARG, "null argument");
    CtxLease L;
    ScStaged st;
    if (!(flags & NMX_SCALARS_DEVICE)) {
      const size_t n = (size_t)1 << num_rounds;
      A = st.up(*L.c, A, n), B = st.up(*L.c, B, n), C = st.up(*L.c, C, n);
      flags |= NMX_SCALARS_DEVICE;
    }
    fv_sumcheck_prove(*L.c, field, 3, claim, taus, num_rounds, A, B, C, flags, transcript, ctx, out_polys, out_r, out_claims);
  });
}
int nmx_sumcheck_prove_quad_prod(int field, const void* claim, size_t num_rounds, void* A, void* B, uint32_t flags,
                                 nmx_transcript_fn transcript, void* ctx, uint8_t* out_polys, uint8_t* out_r, uint8_t* out_claims) {
  return guarded([&] {
    check_sc_args(field, num_rounds, flags, transcript);
    require(claim && A && B, NMX_E_ARG, "null argument");
    CtxLease L;
    ScStaged st;
    if (!(flags & NMX_SCALARS_DEVICE)) {
      const size_t n = (size_t)1 << num_rounds;
      A = st.up(*L.c, A, n), B = st.up(*L.c, B, n);
      flags |= NMX_SCALARS_DEVICE;
    }
    fv_sumcheck_prove(*L.c, field, 4, claim, nullptr, num_rounds, A, B, nullptr, flags, transcript, ctx, out_polys, out_r, out_claims);
  });
}
int nmx_sumcheck_prove_batch_eval(int field, const void* claims, const size_t* num_rounds, void* const* polys, const void* const* eq_points,
                                  const void* coeffs, size_t k, uint32_t flags, nmx_transcript_fn transcript, void* ctx, uint8_t* out_polys,
                                  uint8_t* out_r, uint8_t* out_finals) {
  return guarded([&] {
    check_sc_args(field, 0, flags, transcript);
    require(claims && num_rounds && polys && eq_points && coeffs && k >= 1, NMX_E_ARG, "null argument");
    for (size_t i = 0; i < k; i++) require(polys[i] && eq_points[i] && num_rounds[i] < 31, NMX_E_ARG, "null polynomial / evaluation point");
    CtxLease L;
    ScStaged st;
    std::vector<void*> staged;
    if (!(flags & NMX_SCALARS_DEVICE)) {
      for (size_t i = 0; i < k; i++) staged.push_back(st.up(*L.c, polys[i], (size_t)1 << num_rounds[i]));
      polys = staged.data();
      flags |= NMX_SCALARS_DEVICE;
    }
    fv_sumcheck_prove_batch(*L.c, field, (const uint8_t*)claims, num_rounds, polys, (const uint8_t* const*)eq_points, (const uint8_t*)coeffs, k,
                            flags, transcript, ctx, out_polys, out_r, out_finals);
  });
}

// InnerProductArgument::prove (src/provider/ipa_pc.rs:174-281) as one call.  The commitment key is never folded (ipa.hpp): every
// round's L and R are one fused two-vector commitment over the registered key, whose window tables are the ones every other
// commitment of that curve uses.  Per round: k_ipa_expand (+ the one-block partial sum) -> 64 bytes to the host (c_L, c_R) ->
// commit_batch (the blinding terms c_L * ck_c, c_R * ck_c on pool threads under the MSM) -> the transcript callback -> k_ipa_fold.
int nmx_ipa_prove(uint64_t ck_handle, const void* ck_c_xy64, const void* a, const void* b, size_t n, uint32_t flags,
                  nmx_ipa_transcript_fn transcript, void* ctx, uint8_t* out_L, uint8_t* out_R, uint8_t* out_is_inf, uint8_t* out_a_hat) {
  return guarded([&] {
    require(transcript != nullptr, NMX_E_ARG, "null transcript callback");
    require(ck_c_xy64 && a && b && out_a_hat, NMX_E_ARG, "null argument");
    require(!(flags & ~(uint32_t)(NMX_SCALARS_MONT | NMX_SCALARS_DEVICE | NMX_BASES_MONT)), NMX_E_ARG, "unsupported flag");
    require(n >= 1 && (n & (n - 1)) == 0, NMX_E_ARG, "n must be a power of two");  // U.b_vec.len().ilog2() rounds (ipa_pc.rs:262)
    require(n < ((size_t)1 << 31), NMX_E_TOO_LARGE, "vector too long");
    size_t rounds = 0;
    while (((size_t)1 << rounds) < n) rounds++;
    require(rounds == 0 || (out_L && out_R), NMX_E_ARG, "null argument");
    auto bs = lookup(ck_handle);
    require(n <= bs->n, NMX_E_HANDLE, "ck shorter than the vectors");  // ck.split_at(U.b_vec.len()) would panic (ipa_pc.rs:183)
    const CurveOps& o = ops(bs->curve);
    const int field = o.scalar_field;
    const bool dev = (flags & NMX_SCALARS_DEVICE) != 0;
    const uint32_t sflags = flags & NMX_SCALARS_MONT;
    CtxLease L;
    Ctx& c = *L.c;
    if (rounds == 0) {  // a_hat = a_vec[0], no round (:268-272)
      if (dev) {
        HIPCHK(hipMemcpyAsync(out_a_hat, a, 32, hipMemcpyDeviceToHost, c.stream));
        stream_wait(c.stream);
      } else {
        memcpy(out_a_hat, a, 32);
      }
      return;
    }
    // the call's device state, outside the arena the MSMs re-carve: [a b staged] A0 A1 B0 B1 vL vR S0 S1 dout
    auto pad = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t half = pad((n / 2) * 32), full = pad(n * 32);
    const size_t total = (dev ? 0 : 2 * full) + 4 * half + 2 * full + 2 * half + 256;
    struct Own {
      void* p = nullptr;
      hipEvent_t ev = nullptr;
      ~Own() {
        if (p) (void)hipFree(p);
        if (ev) (void)hipEventDestroy(ev);
      }
    } own;
    char* base;
    if (total <= ((size_t)64 << 20)) {
      aux_reserve(c, total);
      base = c.aux;
    } else {
      HIPCHK(hipMalloc(&own.p, total));
      base = (char*)own.p;
    }
    HIPCHK(hipEventCreateWithFlags(&own.ev, hipEventDisableTiming));
    size_t used = 0;
    auto carve = [&](size_t bytes) {
      char* q = base + used;
      used += bytes;
      return (uint32_t*)q;
    };
    const uint32_t *a_cur, *b_cur;
    if (dev) {
      a_cur = (const uint32_t*)a, b_cur = (const uint32_t*)b;
    } else {
      uint32_t *sa = carve(full), *sb = carve(full);
      HIPCHK(hipMemcpyAsync(sa, a, n * 32, hipMemcpyHostToDevice, c.stream));
      HIPCHK(hipMemcpyAsync(sb, b, n * 32, hipMemcpyHostToDevice, c.stream));
      a_cur = sa, b_cur = sb;
    }
    uint32_t* A[2] = {carve(half), carve(half)};
    uint32_t* B[2] = {carve(half), carve(half)};
    uint32_t *vL = carve(full), *vR = carve(full);
    uint32_t* S[2] = {carve(half), carve(half)};
    uint32_t* dout = carve(256);
    try {
      const bool sharded = !bs->parts.empty();
      const BaseSet& key = sharded ? *bs : prefix_or_key(*bs, 0, n);
      const uint32_t mflags = (flags & (NMX_SCALARS_MONT | NMX_BASES_MONT)) | NMX_SCALARS_DEVICE;
      const int hip_dev = hip_device_of(c.dev);
      uint8_t r[32], rinv[32];
      size_t len = n;
      fv_ipa_one(c, field, S[0]);  // S_0 = [1]
      for (size_t k = 0; k < rounds; k++, len /= 2) {
        // round k's launch folds with round k - 1's challenge on the way: vectors and table ping-pong between two buffers
        const uint32_t* partial_host = nullptr;
        fv_ipa_round(c, field, a_cur, b_cur, S[(k + 1) & 1], A[k & 1], B[k & 1], S[k & 1], n, len, k ? r : nullptr, k ? rinv : nullptr, sflags, vL,
                     vR, own.ev, &partial_host);
        if (k) a_cur = A[k & 1], b_cur = B[k & 1];
        // c_L, c_R reach the host on the pool threads that need them (the blinding terms), once the launch above has completed
        const hipEvent_t ev = own.ev;
        const std::function<void(size_t, uint8_t*)> late = [=](size_t j, uint8_t* out32) {
          HIPCHK(hipSetDevice(hip_dev));
          HIPCHK(hipEventSynchronize(ev));
          fv_ipa_scalar(field, partial_host, n, (int)j, sflags, out32);
        };
        uint8_t pts[128], infs[2] = {0, 0};
        if (!sharded) {
          const BatchItem items[2] = {{vL, n}, {vR, n}};
          stat_add(NMX_STAT_MSM_CALLS, 2);
          o.commit_batch(c, key, items, 2, field_call(nullptr, mflags), ck_c_xy64, nullptr, &late, mflags, pts, infs);
        } else {  // a key over several devices: two sharded commitments (the vectors sit on the primary device)
          uint8_t cs[64];
          late(0, cs), late(1, cs + 32);
          commit_impl(L, *bs, field_call(vL, mflags), n, ck_c_xy64, cs, mflags, pts, infs);
          commit_impl(L, *bs, field_call(vR, mflags), n, ck_c_xy64, cs + 32, mflags, pts + 64, infs + 1);
        }
        memcpy(out_L + 64 * k, pts, 64);
        memcpy(out_R + 64 * k, pts + 64, 64);
        if (out_is_inf) out_is_inf[2 * k] = infs[0], out_is_inf[2 * k + 1] = infs[1];
        require(transcript(ctx, pts, infs[0], pts + 64, infs[1], r) == 0, NMX_E_ARG, "the transcript callback failed");
        require(fv_ipa_invert(field, r, sflags, rinv), NMX_E_ZERO, "a round challenge is zero");  // r.invert().unwrap() (:235)
      }
      fv_ipa_last(c, field, a_cur, r, rinv, sflags, dout, out_a_hat);  // the last round's vector has two elements
    } catch (...) {
      (void)hipStreamSynchronize(c.stream);  // nothing of this call still reads the caller's vectors
      throw;
    }
  });
}

int nmx_r1cs_cross_term(uint64_t hA, uint64_t hB, uint64_t hC, const void* z1, const void* z2, size_t z_len, const void* e,
                        const void* u, uint32_t flags, void* out) {
  return guarded([&] {
    require(z1 && e && u && out, NMX_E_ARG, "null argument");
    require(flags & NMX_SCALARS_DEVICE, NMX_E_ARG, "nmx_r1cs_cross_term works on HBM-resident vectors");
    std::shared_ptr<Global::SparseSet> sp[3];
    {
      std::lock_guard<std::mutex> lk(G.mu);
      const uint64_t hs[3] = {hA, hB, hC};
      for (int j = 0; j < 3; j++) {
        auto it = G.sparse.find(hs[j]);
        if (it == G.sparse.end()) throw Fail{NMX_E_HANDLE, "unknown matrix handle"};
        sp[j] = it->second;
      }
    }
    for (int j = 0; j < 3; j++)
      require(sp[j]->field == sp[0]->field && sp[j]->rows == sp[0]->rows && sp[j]->cols == sp[0]->cols, NMX_E_ARG,
              "A, B, C must share field and shape");
    require(z_len == sp[0]->cols, NMX_E_ARG, "invalid shape");
    if (sp[0]->rows == 0) return;
    const uint32_t* ip[3] = {sp[0]->indptr, sp[1]->indptr, sp[2]->indptr};
    const uint32_t* ix[3] = {sp[0]->indices, sp[1]->indices, sp[2]->indices};
    const uint32_t* dt[3] = {sp[0]->data, sp[1]->data, sp[2]->data};
    CtxLease L;
    fv_r1cs_cross_term(*L.c, sp[0]->field, ip, ix, dt, sp[0]->rows, sp[0]->cols, z1, z2, e, u, flags, out);
  });
}

int nmx_nifs_fold(int field, const void* w1, const void* w2, size_t n_w, const void* e1, const void* t, size_t n_e, const void* r,
                  uint32_t flags, void* w, void* e) {
  return guarded([&] {
    require(((w1 && w2 && w) || n_w == 0) && ((e1 && t && e) || n_e == 0) && r, NMX_E_ARG, "null argument");
    require(flags & NMX_SCALARS_DEVICE, NMX_E_ARG, "nmx_nifs_fold works on HBM-resident vectors");
    require(n_w + n_e < (1ull << 31), NMX_E_TOO_LARGE, "vectors too long");
    if (n_w + n_e == 0) return;
    CtxLease L;
    fv_nifs_fold(*L.c, field, w1, w2, n_w, e1, t, n_e, r, flags, w, e);
  });
}

int nmx_spmv_apply_pair(uint64_t handle, const void* z1, const void* z2, size_t z_len, uint32_t flags, void* out1,
                        void* out2) {
  return guarded([&] {
    require(z1 && z2 && out1 && out2, NMX_E_ARG, "null argument");
    std::shared_ptr<Global::SparseSet> sp;
    {
      std::lock_guard<std::mutex> lk(G.mu);
      auto it = G.sparse.find(handle);
      if (it == G.sparse.end()) throw Fail{NMX_E_HANDLE, "unknown matrix handle"};
      sp = it->second;
    }
    const Global::SparseSet& ss = *sp;
    require(z_len == ss.cols, NMX_E_ARG, "invalid shape for v1 / v2");  // sparse.rs:217-218
    if (ss.rows == 0) return;
    CtxLease L;
    fv_spmv_apply_pair(*L.c, ss.field, ss.indptr, ss.indices, ss.data, ss.rows, ss.cols, z1, z2, flags, out1, out2);
  });
}

int nmx_set_profiling(int on) {
  G.profiling.store(on != 0);
  return NMX_OK;
}
int nmx_profile_last(float* ms, int cap) {
  int n = t_prof_n < cap ? t_prof_n : cap;
  for (int i = 0; i < n; i++) ms[i] = t_prof[i];
  return t_prof_n;
}
int nmx_set_window_bits(uint32_t c) {
  if (c > 24) return NMX_E_ARG;
  G.force_c.store(c);
  return NMX_OK;
}

int nmx_set_option(const char* name, uint32_t value) {
  return guarded([&] {
    require(name != nullptr, NMX_E_ARG, "null argument");
    ensure_init();  // the environment defaults are applied first, then overridden
    const std::string n(name);
    std::lock_guard<std::mutex> lk(G.mu);
    if (n == "no_partition") G.no_partition = value;
    else if (n == "seg_min_total") G.seg_min_total = value;
    else if (n == "seg_min_len") G.seg_min_len = value ? value : 1u;
    else if (n == "seg_lanes") G.seg_lanes_override = value;
    else if (n == "no_quad_accum") G.no_quad_accum = value;
    else if (n == "no_quad_final") G.no_quad_final = value;
    else if (n == "small_blocks") G.small_blocks = value;
    else if (n == "quad_final_below") G.quad_final_below = value;
    else if (n == "accum_prefetch") G.accum_prefetch = value;
    else if (n == "no_batch_fuse") G.no_batch_fuse = value;
    else if (n == "prefix_tables") G.prefix_tables = value > 2 ? 2u : (uint32_t)value;
    else if (n == "no_tree_fuse") G.no_tree_fuse = value;
    else if (n == "tree_threads") G.tree_threads = value;
    else if (n == "big_slice") G.big_slice = value;
    else if (n == "big_threads") G.big_threads = value;
    else if (n == "hist_grid") G.hist_grid = value;
    else if (n == "hist_bs") G.hist_bs = value;
    else if (n == "sync_spin_us") G.sync_spin_us = value;
    else if (n == "sc_poll_us") G.sc_poll_us = value;
    else if (n == "host_split") {
      require(value <= 16 || value == 255, NMX_E_ARG, "host_split: pieces a large host-scalar call is cut into, 0 / 1 = off, at most 16; 255 = by size");
      G.host_split = value;
    } else if (n == "host_split_min_n") G.host_split_min_n = value;
    else if (n == "sc_fused_sum") G.sc_fused_sum = value ? 1u : 0u;
    else if (n == "sc_side_streams") G.sc_side_streams = value ? 1u : 0u;
    else if (n == "sc_quad") G.sc_quad = value ? 1u : 0u;
    else if (n == "sc_prelaunch") G.sc_prelaunch = value ? 1u : 0u;
    else if (n == "sc_host_parts") G.sc_host_parts = value ? 1u : 0u;
    else if (n == "sc_resident") G.sc_resident = value ? 1u : 0u;
    else if (n == "sc_torn_test") G.sc_torn_test = value > 1000 ? 1000u : (uint32_t)value;
    else if (n == "sc_host_tail") {
      require(value <= 8, NMX_E_ARG, "sc_host_tail: log2 of the table length the host takes over, 0..8");
      G.sc_host_tail = value;
    }
    else if (n == "horner_order") G.horner_order = value ? 1u : 0u;

    else if (n == "shard_min_n") G.shard_min_n.store(value ? value : 1u);
    else if (n == "cache_table_after") {  // slice cache: window tables from the (value + 1)-th use of an array on (0: at upload)
      std::lock_guard<std::mutex> ck(SC.mu);
      SC.table_after_uses = value;
    } else if (n == "max_table_mib") G.max_table_bytes.store((size_t)value << 20);  // 0: no limit but the HBM itself
    else if (n == "horner_top") G.horner_top = value;
    else if (n == "horner_window") G.horner_window = value ? value : 64u;
    else if (n == "horner_sub") G.horner_sub = value;
    else if (n == "eq_max_blocks") G.eq_max_blocks = value;
    else if (n == "horner_spin_limit") G.horner_spin_limit = value;
    else if (n == "seg_heavy_above") G.seg_heavy_above = value > 63u ? 63u : value;
    else if (n == "force_peer_copy") G.force_peer_copy = value;
    else if (n == "combine") {
      require(value <= 2, NMX_E_ARG, "combine: 0 auto, 1 host sum, 2 RCCL required");
      G.combine_mode = value;
    } else if (n == "cache_verify") {
      require(value <= 1, NMX_E_ARG, "cache_verify: 0 full, 1 rolling window");
      G.cache_verify = value;
    }
    else throw Fail{NMX_E_ARG, "unknown option name"};
  });
}

int nmx_cache_clear(void) {
  return guarded([&] {
    std::lock_guard<std::mutex> up(SC.upload_mu);
    std::list<SliceEntry> drop;
    {
      std::lock_guard<std::mutex> lk(SC.mu);
      drop.swap(SC.entries);
      SC.bytes = 0;
      cache_publish_gauges();
    }
  });
}
int nmx_cache_invalidate(const void* bases) {
  return guarded([&] {
    std::lock_guard<std::mutex> lk(SC.mu);
    const uint8_t* b = (const uint8_t*)bases;
    for (auto it = SC.entries.begin(); it != SC.entries.end();) {
      auto cur = it++;
      if (b >= cur->host && b < cur->host + 64 * cur->n) cache_erase(cur);
    }
    cache_publish_gauges();
  });
}
int nmx_cache_configure(size_t max_bytes, size_t min_n, size_t max_entries) {
  return guarded([&] {
    std::lock_guard<std::mutex> lk(SC.mu);
    if (max_bytes) SC.max_bytes = max_bytes;
    if (min_n) SC.min_n = min_n;
    if (max_entries) SC.max_entries = max_entries;
  });
}
size_t nmx_min_gpu_n(int curve) {
  (void)curve;  // one threshold for the four curves: the floor is launch + dependent-addition latency, not field size
  if (const char* t = getenv("NMX_MIN_N")) return (size_t)atoll(t);
  return 128;
}
int nmx_check_layout(int curve, const void* generator_raw64, const void* scalar_raw32, uint64_t value) {
  return guarded([&] {
    require(generator_raw64 && scalar_raw32, NMX_E_ARG, "null argument");
    require(ops(curve).check_layout((const uint8_t*)generator_raw64, (const uint8_t*)scalar_raw32, value), NMX_E_FORMAT,
            "in-memory layout is not 4 x u64 little-endian Montgomery (R = 2^256) limbs: use the canonical byte forms");
  });
}
int nmx_stats(uint64_t* out, int cap) {
  g_stats[NMX_STAT_SC_TORN_INJECTED].store(G.sc_torn_injected.load(std::memory_order_relaxed), std::memory_order_relaxed);
  g_stats[NMX_STAT_SC_TORN_REJECTS].store(G.sc_torn_rejects.load(std::memory_order_relaxed), std::memory_order_relaxed);
  for (int i = 0; i < NMX_STAT_COUNT && i < cap; i++) out[i] = g_stats[i].load(std::memory_order_relaxed);
  return NMX_STAT_COUNT;
}

}  // extern "C"
