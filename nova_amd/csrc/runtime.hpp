// runtime.hpp -- device runtime shared by every translation unit of libnova_mi355x.so: error plumbing, the generic
// launch trampoline, the per-call context (HIP stream + workspace arena + profiling events), the key registry
// and the per-curve operation table.  The library is split into one TU per curve (compile time), one for the
// rocPRIM sort and one for the C ABI (capi.hip).  There is no CPU fallback anywhere: without a HIP device every
// entry point returns NMX_E_NO_DEVICE.
#pragma once
#include <functional>
#include <type_traits>
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <exception>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/nova_mi355x.h"
#include "curves.hpp"
#include "msm_pipeline.hpp"
#include "curve_quad.hpp"

namespace nmx {

// ---------------------------------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------------------------------
struct Fail {
  int code;
  std::string msg;
};
#define HIPCHK(x)                                                                                  \
  do {                                                                                             \
    hipError_t e_ = (x);                                                                           \
    if (e_ != hipSuccess)                                                                          \
      throw ::nmx::Fail{NMX_E_HIP, std::string(#x) + ": " + hipGetErrorString(e_)};                \
  } while (0)
static inline void require(bool ok, int code, const char* msg) {
  if (!ok) throw Fail{code, msg};
}

// ---------------------------------------------------------------------------------------------------
// generic launch trampoline: one lane per tid
// ---------------------------------------------------------------------------------------------------
// functors with `static constexpr bool kFullWaves` use wave-wide operations: every lane of the block calls them,
// lanes past n with valid = false
template <class F, class = void> struct wants_full_waves : std::false_type {};
template <class F> struct wants_full_waves<F, std::void_t<decltype(F::kFullWaves)>> : std::true_type {};
template <class F> __global__ __launch_bounds__(256) void k_launch(F f, uint32_t n) {
  uint32_t tid = blockIdx.x * 256u + threadIdx.x;
  if constexpr (wants_full_waves<F>::value) f(tid, tid < n);
  else if (tid < n) f(tid);
}

// ---------------------------------------------------------------------------------------------------
// per-call context: stream + workspace arena + profiling events
// ---------------------------------------------------------------------------------------------------
static constexpr int kMaxMarks = 12;
struct Ctx {
  int dev = 0;  // logical device (index into Global::hip_dev) the stream, the arenas and the events belong to
  hipStream_t stream = nullptr;
  char* arena = nullptr;
  size_t cap = 0;
  char* pinned = nullptr;  // 64 KiB of pinned host memory: landing zone of the small device->host result copies
  char* aux = nullptr;  // second, small arena: data that must outlive calls which re-carve `arena` (eq tables)
  size_t aux_cap = 0;
  // mailbox of the sum-check provers (sumcheck_prove.hpp): coherent pinned host memory the round kernels write their sums
  // into, sequence word last, and the host polls -- mail = host address, mail_dev = the same bytes as the device sees them
  char *mail = nullptr, *mail_dev = nullptr;
  uint32_t mail_seq = 0;
  // side streams of the batch prover (sumcheck_prove.hpp): the claims of a round are independent passes, one stream each
  static constexpr int kSideStreams = 15;
  hipStream_t side[kSideStreams] = {};
  hipEvent_t side_ev = nullptr;
  // challenge lines of the provers' pre-launched passes: uncached device memory the HOST writes through the large BAR (one 256-byte
  // stride per mailbox slot), nullptr when the device has no large BAR (then nothing is pre-launched)
  uint32_t* chal = nullptr;
  bool chal_tried = false;
  hipEvent_t ev[kMaxMarks];
  bool have_ev = false;
  hipEvent_t async_ev = nullptr;  // behind the last NMX_ASYNC call enqueued on this context
  std::vector<XYZZW> wsum;  // landing buffer of the per-window sums (off the caller's stack)
  uint64_t shape_key = 0;   // shape of the last MSM sized on this context and the workspace it needs
  size_t shape_bytes = 0;
};

// A key resident in HBM.  Owned through shared_ptr: the registry (or the slice cache) holds one reference and every
// in-flight call holds another, so nmx_bases_unregister / a cache eviction on one host thread can never free the
// device memory under an MSM running on another (the trait is called from rayon workers).
struct BaseSet {
  int curve = 0;
  size_t n = 0;
  int dev = 0;         // logical device `d` lives on (Global::hip_dev)
  void* d = nullptr;   // AffineW[pre_W ? pre_W * n : n]: the key (internal form), then its window tables
  // A key sharded over the devices of the process (nmx_init_devices): d == nullptr and parts[i] is an
  // ordinary single-device key holding points [part_begin[i], part_begin[i + 1]) with its own window tables.
  std::vector<std::shared_ptr<BaseSet>> parts;
  std::vector<size_t> part_begin;  // parts.size() + 1 entries
  uint32_t pre_c = 0;  // window width of the tables (0: none)
  uint32_t pre_W = 0;
  bool any_identity = true;  // false: no point of the key is the identity (checked at registration): the digit stage of
                             // an MSM then never reads the bases (64 B per pair saved); true is always safe
  bool owns = true;    // false: `d` belongs to somebody else (one-shot uploads wrapped for batch_impl)
  BaseSet() = default;
  BaseSet(int curve_, size_t n_) : curve(curve_), n(n_) {}
  BaseSet(const BaseSet&) = delete;
  BaseSet& operator=(const BaseSet&) = delete;
  size_t alloc_bytes = 0;  // bytes behind `d` (0: derive from pre_W -- wrapped one-shot uploads)
  // Keys whose tables are wider than 17 bits (>= 2^22 points: c = 20, 2^19 buckets per set) also keep a SECOND, narrower table
  // set over their first 2^18 points (round 4): an MSM or a batch that stays inside that prefix -- HyperKZG's batch_commit of
  // n/2 ... 2 over a 2^22+ key, src/provider/hyperkzg.rs:593-612,1100 -- runs on it: 2^15 buckets to reduce instead of 2^19,
  // and up to 32 vectors fused per run where the wide key takes none.  An ordinary key object of its own (same device).
  std::shared_ptr<BaseSet> prefix;
  size_t bytes() const {
    size_t b = d ? (alloc_bytes ? alloc_bytes : n * 64 * (pre_W ? pre_W : 1)) : 0;
    for (const auto& p : parts) b += p->bytes();
    if (prefix) b += prefix->bytes();
    return b;
  }
  ~BaseSet();  // capi.hip: hipFree(d) on its device
};
using BaseRef = std::shared_ptr<const BaseSet>;

// A registered key gets window tables from this many points on, and MSMs over it use them from this many pairs on:
// everything but the trivial sizes.  (4096 until the small sizes were measured, scripts/gpu_smallmsm.py: with tables a
// 2^11-pair MSM takes 0.29 ms instead of 0.98 -- one bucket set instead of 37, no 250-doubling Horner on the host.)
static constexpr size_t kPrecompMinN = 2;  // default of Global::precomp_min_n
// internal upload flag (never part of the ABI): the source array is a resident key already in the internal form -- no
// ---------------------------------------------------------------------------------------------------
// Helper threads: the shards of a multi-device call, the lanes of a batch, the host-side work that runs under a GPU MSM (the
// rolling verification of a cached slice, the blinding term of a commitment).  Persistent: creating and joining a std::thread
// costs 20-60 us -- seven of them per MSM on an 8-GPU node, one per slice-form call.  A worker runs one job at a time and goes
// back to the idle list; the pool grows to the largest number of helpers ever wanted at once and lives as long as the process
// (leaked on purpose, like G: its threads may still be parked when static destructors run).
// ---------------------------------------------------------------------------------------------------
struct Worker {
  std::thread th;
  std::mutex m;
  std::condition_variable cv;
  std::function<void()> job;
  bool has_job = false;
};
struct WorkerPool {
  std::mutex mu;
  std::vector<Worker*> idle;
  static void loop(Worker* w) {
    for (;;) {
      std::function<void()> j;
      {
        std::unique_lock<std::mutex> lk(w->m);
        w->cv.wait(lk, [w] { return w->has_job; });
        j = std::move(w->job);
        w->has_job = false;
      }
      j();  // must not throw; ends by handing the worker back (release)
    }
  }
  Worker* acquire() {  // may throw std::system_error (thread creation)
    {
      std::lock_guard<std::mutex> lk(mu);
      if (!idle.empty()) {
        Worker* w = idle.back();
        idle.pop_back();
        return w;
      }
    }
    Worker* w = new Worker;
    try {
      w->th = std::thread(loop, w);
    } catch (...) {
      delete w;
      throw;
    }
    return w;
  }
  void release(Worker* w) {
    std::lock_guard<std::mutex> lk(mu);
    idle.push_back(w);
  }
  void submit(Worker* w, std::function<void()> j) {
    {
      std::lock_guard<std::mutex> lk(w->m);
      w->job = std::move(j);
      w->has_job = true;
    }
    w->cv.notify_one();
  }
};
WorkerPool& worker_pool();  // capi.hip

// f() on a pool worker while the caller does something else; get() waits and rethrows.  Like the future of std::async, the
// destructor waits (the task may refer to the caller's frame).  Without a thread to run on, f() runs here and now.
template <class R> class PoolFuture {
  struct State {
    std::mutex m;
    std::condition_variable cv;
    bool done = false;
    R value{};
    std::exception_ptr err;
  };
  std::shared_ptr<State> st;
  void wait() {
    std::unique_lock<std::mutex> lk(st->m);
    st->cv.wait(lk, [this] { return st->done; });
  }

 public:
  PoolFuture() = default;
  template <class F> explicit PoolFuture(F f) : st(std::make_shared<State>()) {
    std::shared_ptr<State> s = st;
    auto body = [s, f]() mutable {
      try {
        s->value = f();
      } catch (...) {
        s->err = std::current_exception();
      }
    };
    WorkerPool& wp = worker_pool();
    Worker* w = nullptr;
    try {
      w = wp.acquire();
      wp.submit(w, [s, body, w, &wp]() mutable {
        body();
        wp.release(w);
        std::lock_guard<std::mutex> lk(s->m);
        s->done = true;
        s->cv.notify_all();
      });
    } catch (const std::exception&) {  // no thread (or no memory for the job): compute inline
      if (w) wp.release(w);
      body();
      st->done = true;
    }
  }
  PoolFuture(const PoolFuture&) = delete;
  PoolFuture& operator=(const PoolFuture&) = delete;
  PoolFuture(PoolFuture&& o) noexcept : st(std::move(o.st)) {}
  PoolFuture& operator=(PoolFuture&& o) noexcept {
    if (st) wait();
    st = std::move(o.st);
    return *this;
  }
  ~PoolFuture() {
    if (st) wait();
  }
  bool valid() const { return st != nullptr; }
  R get() {
    wait();
    std::shared_ptr<State> s = std::move(st);
    if (s->err) std::rethrow_exception(s->err);
    return std::move(s->value);
  }
};

// conversion, no validation (the slice cache adding window tables to a key it holds)
static constexpr uint32_t NMX_BASES_INTERNAL = 1u << 30;
struct Global {
  std::mutex mu;
  bool inited = false;
  int device = 0;  // HIP device of logical device 0 (the primary: field-vector kernels, unsharded keys, device scalars)
  // Logical devices of this process (nmx_init_devices): hip_dev[i] = HIP device ordinal.  Entries are only ever appended
  // (capacity reserved at start-up, so readers never see a reallocation); ndev_active of them receive the shards of
  // newly registered keys.  Logical devices may share a physical GPU (NMX_DEVICES_OVERSUBSCRIBE: tests on a 1-GPU box).
  static constexpr size_t kMaxDevices = 64;
  std::vector<int> hip_dev;
  std::atomic<uint32_t> ndev_active{1};
  std::atomic<size_t> shard_min_n{(size_t)1 << 20};  // keys shorter than this stay whole on the primary device
  std::atomic<size_t> max_table_bytes{0};  // option max_table_mib / env NMX_MAX_TABLE_MIB: window tables larger than this are not built (0: no limit)
  std::vector<std::vector<Ctx*>> free_ctx;  // per logical device
  std::vector<Ctx*> all_ctx;
  std::unordered_map<uint64_t, std::shared_ptr<BaseSet>> bases;
  struct SparseSet {  // a CSR matrix resident in HBM (R1CS matrices are fixed per circuit: upload once)
    int field = 0;
    size_t rows = 0, cols = 0, nnz = 0;
    uint32_t *indptr = nullptr, *indices = nullptr, *data = nullptr;
    // M^T for compute_eval_table_sparse (nmx_spmv_apply_transposed), built from the resident CSR on first use: the CSC arrays
    // cut into VIRTUAL rows (a column of an R1CS matrix can hold one entry per constraint -- the constant-one column does --
    // so a column longer than 32 entries is split into chunks of 16 entries, each its own lane; a block sums a column's partials), the slot
    // every virtual row writes (an output row, or 2^31 | index of a partial), and the (row, first partial) list of the split rows
    struct Transposed {
      uint32_t *vptr = nullptr, *indices = nullptr, *data = nullptr, *vout = nullptr, *hrow = nullptr, *hstart = nullptr;
      size_t nvirt = 0, nheavy = 0, nparts = 0;
      int dev = 0;
      ~Transposed();  // capi.hip
    };
    std::mutex t_mu;
    std::shared_ptr<Transposed> tr;
    SparseSet() = default;
    SparseSet(const SparseSet&) = delete;
    SparseSet& operator=(const SparseSet&) = delete;
    ~SparseSet();  // capi.hip
  };
  std::unordered_map<uint64_t, std::shared_ptr<SparseSet>> sparse;
  uint64_t next_handle = 1;
  // written by nmx_set_profiling / nmx_set_window_bits while calls on other threads read them
  std::atomic<bool> profiling{false};
  std::atomic<uint32_t> force_c{0};
  std::atomic<uint32_t> force_lmax{0};  // env NMX_TUNE_LMAX (tuning only)
  std::atomic<size_t> precomp_min_n{kPrecompMinN};  // env NMX_TUNE_PRECOMP_MIN_N (tuning only)
  std::atomic<uint32_t> force_fold_t{0};  // env NMX_TUNE_FOLD_T (tuning only)
  std::atomic<uint32_t> no_quad_accum{0};  // env NMX_TUNE_NO_QUAD_ACCUM (tuning only)
  std::atomic<uint32_t> no_partition{0};   // env NMX_TUNE_NO_PARTITION: generic radix-sort path everywhere (A/B runs)
  std::atomic<uint32_t> seg_min_total{kSegMinTotalAuto};  // env NMX_TUNE_SEG_MIN_TOTAL (profiles/r02_msm_2p20/seg_threshold.txt): msm_seg.hpp from this many sorted entries (0xffffffff: never)
  std::atomic<uint32_t> seg_min_len{8};           // env NMX_TUNE_SEG_MIN_LEN
  std::atomic<uint32_t> seg_lanes_override{0};    // env NMX_TUNE_SEG_LANES (0: the kernel's resident lane count)
  std::atomic<uint32_t> small_blocks{8};          // option small_blocks / env NMX_TUNE_SMALL_BLOCKS: bucket sums of MSMs with <= 1024 buckets in two block-level launches (curve_quad.hpp k_small_accum), this many entries per quad; 0: the task path (plan, expand, accumulate, strided folds)
  std::atomic<uint32_t> no_quad_final{0};         // env NMX_TUNE_NO_QUAD_FINAL
  std::atomic<uint32_t> quad_final_below{65536};  // env NMX_TUNE_QUAD_FINAL_BELOW / option quad_final_below: the final pass runs four lanes per bucket below this many buckets
  std::atomic<uint32_t> accum_prefetch{0};        // env NMX_TUNE_ACCUM_PF / option accum_prefetch: 0 = by table size, 1, 2
  std::atomic<uint32_t> horner_top{0};            // env NMX_TUNE_HORNER_TOP / option horner_top: suffix Horner from 1024 coefficients: 0 = single-pass scan (k_horner_scan); the two-pass kernels: 8 = 8-element chunks in registers, 4, 1 = chunk-per-lane recursion only
  std::atomic<uint32_t> eq_max_blocks{0};         // option eq_max_blocks: grid cap of the eq-factored sum passes (0 = 2048)
  std::atomic<uint32_t> horner_sub{0};            // option horner_sub: 512-coefficient sub-tiles per wave of the single-pass scan (0 = by size, 1, 2, 4)
  std::atomic<uint32_t> horner_spin_limit{0};     // option horner_spin_limit: polls before a wave of the scan gives up (0 = 2^22; tests set 1 to force the fall-back)
  std::atomic<uint32_t> horner_window{64};        // option horner_window: tiles per look-back round of the single-pass scan (tests: 1 .. 63 force the multi-round path)
  std::atomic<uint32_t> seg_heavy_above{0};       // env NMX_TUNE_SEG_HEAVY_ABOVE / option seg_heavy_above: 0 = by pieces per bucket (8 or 12)
  std::atomic<uint32_t> prefix_tables{2};         // env NMX_TUNE_PREFIX_TABLES / option prefix_tables: narrower table sets over a key's first points (capi.hip add_prefix_tables): 0 none, 1 the 2^18-point set of wide-table keys only, 2 the whole chain (batches descend it)
  std::atomic<uint32_t> no_batch_fuse{0};         // env NMX_TUNE_NO_BATCH_FUSE / option no_batch_fuse: every vector of a batch runs alone
  std::atomic<uint32_t> big_threads{0};           // env NMX_TUNE_BIG_THREADS / option big_threads: block size of the big-bucket pass (128 default, 256, 512)
  std::atomic<uint32_t> big_slice{0};             // env NMX_TUNE_BIG_SLICE / option big_slice: pieces per block of the big-bucket pass (0 = default)
  std::atomic<uint32_t> tree_threads{0};          // env NMX_TUNE_TREE_THREADS / option tree_threads: block size of the fused reduction tree (0 = default, 256 or 512)
  std::atomic<uint32_t> no_tree_fuse{0};          // env NMX_TUNE_NO_TREE_FUSE / option no_tree_fuse: 0 / 2 = fused reduction tree (default), 1 = one launch per reduction level
  std::atomic<uint32_t> hist_grid{0};             // env NMX_TUNE_HIST_GRID / option hist_grid
  std::atomic<uint32_t> hist_bs{0};               // env NMX_TUNE_HIST_BS / option hist_bs: threads per block of k_hist_hi (0: as k_part_hi)
  std::atomic<uint32_t> horner_order{1};          // option horner_order: 1 = tiles of k_horner_scan by start-order ticket, 0 = by block id
  std::atomic<uint32_t> host_split{255};           // option host_split: a call with HOST scalars over >= host_split_min_n pairs of a single-device key is cut into this many pieces whose uploads overlap the previous piece's MSM (0 / 1: off)
  std::atomic<size_t> host_split_min_n{(size_t)1 << 19};
  std::atomic<uint32_t> sc_fused_sum{1};          // option sc_fused_sum: 1 = a sum-check round is ONE launch (the last block sums the partials, k_sc_pass); 0 = pass + final-sum launch
  std::atomic<uint32_t> sc_side_streams{1};       // option sc_side_streams: the batch prover runs claim i > 0 on its own stream (0: all on the context's)
  std::atomic<uint32_t> sc_prelaunch{1};          // option sc_prelaunch: the cubic / quad_prod provers enqueue their small passes a round early; the pass takes its challenge from pinned memory (0: launched when the challenge is known)
  std::atomic<uint32_t> sc_host_parts{1};         // option sc_host_parts: passes of <= 64 blocks send per-block partial sums to the host, which adds them (0: last-block ticket)
  std::atomic<uint32_t> sc_resident{1};           // option sc_resident: from tables of <= 2^14 elements every remaining device round of a sum-check prover runs inside ONE resident kernel (k_sc_resident) that waits for each challenge on the device (0: a pass per round)
  std::atomic<uint32_t> sc_torn_test{0};          // option sc_torn_test (tests): microseconds a deliberately TORN challenge line stays on the device before the whole one follows (0: off)
  std::atomic<uint64_t> sc_torn_injected{0}, sc_torn_rejects{0};  // NMX_STAT_SC_TORN_INJECTED / _REJECTS
  std::atomic<uint32_t> sc_quad{1};               // option sc_quad: passes of <= 2^12 indices of the cubic / quad_prod provers run four lanes per index (0: one)
  std::atomic<uint32_t> sc_host_tail{7};          // option sc_host_tail: the sum-check provers finish on the host once the tables hold <= 2^this elements (0: only the final values come over; max 8)
  std::atomic<uint32_t> sc_poll_us{2000};         // option sc_poll_us: the sum-check provers poll a round's mailbox this long before they synchronise the stream (0: always synchronise)
  std::atomic<uint32_t> sync_spin_us{0};          // env NMX_SYNC_SPIN_US / option sync_spin_us: poll the stream this long before blocking
  std::atomic<uint32_t> force_peer_copy{0};       // option force_peer_copy: HBM-resident scalars of a sharded call take the staging + hipMemcpyPeerAsync branch even when source and destination are the same GPU (tests on a 1-GPU box)
  std::atomic<uint32_t> combine_mode{0};          // option combine: 0 = RCCL all-gather when the shards sit on >= 2 GPUs, host sum otherwise; 1 = host sum; 2 = RCCL required (also with one GPU: tests)
  std::atomic<uint32_t> cache_verify{0};          // option cache_verify: 0 = every hit re-hashes the caller's whole slice (on pool workers, under the MSM); 1 = rolling window (callers that register immutable keys)
  std::atomic<int32_t> launch_gap_ns{-1};         // cost of one dependent tiny launch on this box, measured once (capi.hip launch_gap_ns)
};
extern Global& G;                // capi.hip (heap singleton, never destroyed)
void note_table_fallback();                // NMX_STAT_TABLE_FALLBACKS (capi.hip)
void note_scan_timeout();                  // NMX_STAT_SCAN_TIMEOUTS
int32_t launch_gap_ns(hipStream_t stream);  // measured once per process (capi.hip)
void prof_store(const float* ms, int n);   // last call's stage times of this thread (capi.hip)
void prof_add_tail(float ms);
void arena_reserve(Ctx& c, size_t bytes);  // capi.hip
void aux_reserve(Ctx& c, size_t bytes);    // capi.hip
// NMX_ASYNC (capi.hip): records an event behind the work just enqueued on c.stream and remembers it for the calling host
// thread; the thread's next lease of ANY context waits for it on that context's stream (CtxLease), nmx_sync waits on the host.
void async_mark(Ctx& c);
// rocPRIM radix sort of (key, value) pairs, its own TU (sort.hip).  tmp == nullptr: size query.
void device_sort_pairs(void* tmp, size_t& tmp_bytes, uint32_t* k_in, uint32_t* k_out, uint32_t* v_in,
                       uint32_t* v_out, size_t total, uint32_t bits, hipStream_t stream);

// End-of-call wait.  hipStreamSynchronize may block on an interrupt (tens of microseconds to wake up: a visible share of a
// 0.3 ms MSM, and of every field-vector call); option sync_spin_us = N polls hipStreamQuery for up to N microseconds first.
static inline void stream_wait(hipStream_t s) {
  const uint32_t spin_us = G.sync_spin_us.load(std::memory_order_relaxed);
  if (spin_us) {
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
      const hipError_t e = hipStreamQuery(s);
      if (e == hipSuccess) return;
      if (e != hipErrorNotReady) break;  // a real error: let hipStreamSynchronize report it
      if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(spin_us)) break;
    }
    (void)hipGetLastError();
  }
  HIPCHK(hipStreamSynchronize(s));
}

struct DeviceBackend {
  Ctx& c;
  bool dry;
  size_t used = 0;
  int nmarks = 0;
  bool prof;
  explicit DeviceBackend(Ctx& ctx, bool dry_, bool prof_) : c(ctx), dry(dry_), prof(prof_) {}

  template <class T> T* alloc(size_t count) {
    size_t bytes = (count * sizeof(T) + 255) & ~(size_t)255;
    T* p = (T*)(c.arena + used);
    used += bytes;
    if (!dry) require(used <= c.cap, NMX_E_HIP, "workspace arena overflow");
    return p;
  }
  void memset0(void* p, size_t bytes) {
    if (dry) return;
    HIPCHK(hipMemsetAsync(p, 0, bytes, c.stream));
  }
  template <class F> void launch(const F& f, uint32_t n) {
    if (dry || n == 0) return;
    hipLaunchKernelGGL((k_launch<F>), dim3((n + 255) / 256), dim3(256), 0, c.stream, f, n);
    HIPCHK(hipGetLastError());
  }
  // block-level kernels (msm_partition.hpp): explicit grid / block, static LDS
  template <class A> void launch_kernel(void (*k)(A), uint32_t grid, uint32_t block, const A& a) {
    if (dry || grid == 0) return;
    hipLaunchKernelGGL(k, dim3(grid), dim3(block), 0, c.stream, a);
    HIPCHK(hipGetLastError());
  }
  // Fold / reduction passes with fewer work items than the chip has lanes are bound by the latency of a point
  // addition: they run with four cooperating lanes per addition (curve_quad.hpp, ~3x shorter latency, ~1.3x the
  // lane-cycles).  Passes with enough items to be throughput-bound keep one lane per addition.
  static constexpr uint32_t kQuadBelowItems = 65536;  // 256 CUs x 4 SIMDs x 64 lanes
  template <int FID>
  void launch_fold(const uint32_t* counters, const HeavyRec* heavy, XYZZW* partials, XYZZW* buckets, uint32_t T,
                   uint32_t cap, uint32_t groups) {
    // T >= 64 passes only ever touch the few buckets that hold a large share of all points
    if (T >= 64 || groups * T < kQuadBelowItems) {
      FoldQuadFn<FID> f{counters, heavy, partials, buckets, T, cap, groups};
      launch(f, groups * T * 4);
    } else {
      FoldFn<FID> f{counters, heavy, partials, buckets, T, cap, groups};
      launch(f, groups * T);
    }
  }
  // accumulate: `tasks` = buckets + split tasks that actually exist (estimate), `slots` = lanes to launch for them
  template <int FID>
  void launch_accum(const AffineW* bases, const uint32_t* vals, const uint32_t* start, const uint32_t* end,
                    const uint32_t* counters, const TaskRec* extra, XYZZW* buckets, XYZZW* partials, const MsmShape& sh,
                    uint32_t slots, uint64_t tasks) {
    // a quad per task pays while 4 x tasks is about the chip's 65536 lanes (measured: 10 k tasks 0.139 -> 0.085 ms,
    // 17 k tasks 0.141 -> 0.127, but 59 k tasks 0.148 -> 0.233)
    if (4 * tasks <= kQuadBelowItems + kQuadBelowItems / 2 && !G.no_quad_accum) {
      AccumQuadFn<FID> f{bases, vals, start, end, counters, extra, buckets, partials, sh};
      launch(f, slots * 4);
    } else {
      AccumFn<FID> f{bases, vals, start, end, counters, extra, buckets, partials, sh};
      launch(f, slots);
    }
  }
  // segment-balanced accumulate (msm_seg.hpp): a small multiple of the lanes the chip holds resident for that kernel.
  // Measured (profiles/r02_msm_2p20/seg_lanes_sweep.txt; 196 608 lanes are resident): whole multiples only -- 1.5x
  // leaves half the chip idle in the second round (+7 %); 1x (every wave in lock step from start to end) accumulate
  // 1.25 ms + fold 0.06 at 2^20, 3x 1.17 + 0.13; at 2^21 3x wins by 2.5 % (3.05 against 3.13 ms).
  // Round 3 (profiles/r03_msm_2p20/seg_lanes_sweep.txt), after the plan step moved into the accumulate kernel and the fold
  // stage shrank: the accumulate kernel gains 1-3 % per extra round of lanes (2^20: 1.091 / 1.050 / 1.028 ms at 1x / 2x / 3x
  // resident), the final pass loses 30 us per round (pieces per bucket 3 / 6 / 9 at c = 17: fold 0.051 / 0.083 / 0.110 ms).
  // 2^20: a wash (1.565-1.576 / 1.559-1.561 / 1.555-1.575 ms, two alternating repetitions on one box); 2^18: one round
  // wins by 4 % (0.687 against 0.719 ms); 2^21: two rounds by 1.6 % (2.986 against 3.036 at 1x, 2.987 at 3x).  Hence one
  // round up to 2^23 sorted entries (keys up to 2^19 points), two above.  Half rounds (1.5x) always lose: the second
  // round runs half empty (2^20: 1.704 ms).
  static uint32_t seg_rounds(size_t total_entries) { return total_entries <= ((size_t)1 << 23) ? 1u : 2u; }
  template <int FID> uint32_t seg_lanes(size_t total_entries) {
    static const uint32_t resident = [] {
      int blocks = 0, dev = 0;
      hipDeviceProp_t prop;
      if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0u;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, k_launch<AccumSegFn<FID, 1>>, 256, 0) != hipSuccess) return 0u;
      return (uint32_t)blocks * 256u * (uint32_t)prop.multiProcessorCount;
    }();
    const uint32_t ov = G.seg_lanes_override.load(std::memory_order_relaxed);
    return ov ? ov : seg_rounds(total_entries) * resident;
  }
  template <int FID>
  void launch_fold_raw(const uint32_t* counters, const HeavyRec* list, XYZZL* partial_raw, uint32_t T, uint32_t cap,
                       uint32_t groups, uint32_t use_big) {
    if (groups * T < kQuadBelowItems) {
      FoldRawQuadFn<FID> f{counters, list, partial_raw, T, cap, groups, use_big};
      launch(f, groups * T * 4);
    } else {
      FoldRawFn<FID> f{counters, list, partial_raw, T, cap, groups, use_big};
      launch(f, groups * T);
    }
  }
  template <int FID>
  void launch_final_seg(const uint32_t* start, const uint32_t* end, const uint32_t* total_p, const XYZZL* bucket_raw,
                        const XYZZL* partial_raw, XYZZW* buckets, uint32_t nbuckets, uint32_t lanes, uint32_t min_seg,
                        uint32_t heavy_above) {
    if (nbuckets < G.quad_final_below.load(std::memory_order_relaxed) && !G.no_quad_final) {
      FinalSegQuadFn<FID> f{start, end, total_p, bucket_raw, partial_raw, buckets, nbuckets, lanes, min_seg, heavy_above};
      launch(f, nbuckets * 4);
    } else {
      FinalSegFn<FID> f{start, end, total_p, bucket_raw, partial_raw, buckets, nbuckets, lanes, min_seg, heavy_above};
      launch(f, nbuckets);
    }
  }
  // small MSMs: the bucket sums in two block-level launches (curve_quad.hpp k_small_accum / k_small_combine); 0: not enabled
  uint32_t small_chunk() const {
    const uint32_t per_quad = G.small_blocks.load(std::memory_order_relaxed);
    return per_quad ? 64u * (per_quad > 64u ? 64u : per_quad) : 0u;
  }
  template <int FID>
  void launch_small_accum(const AffineW* bases, const uint32_t* vals, const uint32_t* start, const uint32_t* end, XYZZW* part, XYZZW* buckets,
                          uint32_t nbuckets, uint32_t chunk, uint32_t blocks) {
    if (dry) return;
    const SmallAccArgs a{bases, vals, start, end, part, buckets, nbuckets, chunk};
    hipLaunchKernelGGL((k_small_accum<FID>), dim3(blocks), dim3(256), 0, c.stream, a);
    HIPCHK(hipGetLastError());
  }
  template <int FID>
  void launch_small_combine(const AffineW* bases, const uint32_t* vals, const uint32_t* start, const uint32_t* end, XYZZW* part,
                            XYZZW* buckets, uint32_t nbuckets, uint32_t chunk) {
    if (dry) return;
    const SmallAccArgs a{bases, vals, start, end, part, buckets, nbuckets, chunk};
    hipLaunchKernelGGL((k_small_combine<FID>), dim3(nbuckets), dim3(256), 0, c.stream, a);
    HIPCHK(hipGetLastError());
  }
  // every big bucket in one launch (curve_quad.hpp k_big_all); a bucket spans at most `lanes` pieces, `big_cap` buckets can be big
  template <int FID>
  void launch_big_all(const uint32_t* counters, const HeavyRec* big, const uint32_t* items, const uint32_t* gbase,
                      const XYZZL* bucket_raw, XYZZL* partial_raw, XYZZW* buckets, uint32_t* done, uint32_t* gdone,
                      uint32_t slice) {
    if (dry) return;
    const BigAllArgs a{counters, big, items, gbase, bucket_raw, partial_raw, buckets, done, gdone, slice};
    const uint32_t bt = G.big_threads.load(std::memory_order_relaxed);  // 512 threads per CU resident either way
    if (bt == 512) hipLaunchKernelGGL((k_big_all<FID, 512>), dim3(256), dim3(512), 0, c.stream, a);
    else if (bt == 256) hipLaunchKernelGGL((k_big_all<FID, 256>), dim3(512), dim3(256), 0, c.stream, a);
    else hipLaunchKernelGGL((k_big_all<FID, 128>), dim3(1024), dim3(128), 0, c.stream, a);
    HIPCHK(hipGetLastError());
  }
  // Bucket reduction sum_k (k + 1) B_k per bucket set: the pair tree of ReducePairFn, two dependent quad additions per level.
  // Levels with more inputs than one round of blocks holds (256 CUs x 128 inputs: the kernel runs one 512-thread block per CU
  // at 181 registers) are throughput-bound and keep one launch each; the others run fused, at most seven levels per launch
  // (k_reduce_tree): 16 levels = 1 + 3 launches at c = 17, 15 = 3 at c = 16, 7 = 1 at c = 8.  Returns the WB sums.
  static constexpr uint32_t kTreeThreads = 512, kTreeLevels = 7, kTreeMaxInputs = 256 * 128;
  template <int FID> const XYZZW* reduce_tree(const XYZZW* buckets, const MsmShape& sh, const uint32_t* err_src, bool* err_appended) {
    const XYZZW* D = buckets;
    const XYZZW* Y = buckets;
    uint32_t n_in = sh.M, first = 1;  // M == 1 (c == 1): the bucket is the window sum
    // Fused is the default (no_tree_fuse = 1 restores one launch per level).  A level is two dependent quad additions, 5.3 us
    // for a wave that has its SIMD to itself (profiles/r03_msm_2p20/add_latency.txt); one launch per level adds the launch
    // gap and the trip through memory (~10 us per level; 15-20 us on the boxes of the pool whose dependent launches are slow:
    // BENCH_r02, reduce 0.341 ms), the fused levels cost 7-8 us since the two roles sit on different SIMDs
    // (tail_ab.txt: 2^20 reduce 0.167 -> 0.152 ms, 2^13 0.067 -> 0.055).
    const uint32_t mode = G.no_tree_fuse.load(std::memory_order_relaxed);
    const bool per_level = mode == 1;
    if (!dry) (void)launch_gap_ns(c.stream);  // diagnostic only (nmx_stats: NMX_STAT_LAUNCH_GAP_NS), measured once per process
    while (n_in > 1 && (per_level || (uint64_t)sh.WB * n_in > kTreeMaxInputs)) {
      const uint32_t half = n_in / 2, pairs = sh.WB * half;
      XYZZW* Do = alloc<XYZZW>(pairs);
      XYZZW* Yo = alloc<XYZZW>(pairs);
      if (!dry) {
        if (2 * pairs < kQuadBelowItems) {
          const uint32_t padded = (pairs + 15u) & ~15u;  // 16 quads = one wave: roles never share a wave
          ReducePairQuadFn<FID> f{D, Y, Do, Yo, n_in, pairs, padded, first};
          launch(f, 2 * padded * 4);
        } else {
          const uint32_t padded = (pairs + 63u) & ~63u;
          ReducePairFn<FID> f{D, Y, Do, Yo, n_in, pairs, padded, first};
          launch(f, 2 * padded);
        }
      }
      D = Do, Y = Yo, n_in = half, first = 0;
    }
    uint32_t levels = 0;
    while ((1u << levels) < n_in) levels++;
    const uint32_t tt = G.tree_threads.load(std::memory_order_relaxed) == 256 ? 256u : kTreeThreads;
    const uint32_t max_lv = tt == 256 ? kTreeLevels - 1 : kTreeLevels;  // a block owns tt / 4 inputs
    const uint32_t launches = (levels + max_lv - 1) / max_lv;
    for (uint32_t i = 0; i < launches; i++) {
      const uint32_t lv = levels / launches + (i < levels % launches ? 1u : 0u);
      const uint32_t n_total = sh.WB * n_in, n_out = n_total >> lv;
      const bool last = i + 1 == launches;
      XYZZW* Do = alloc<XYZZW>(n_out);
      XYZZW* Yo = alloc<XYZZW>(n_out + (last ? 1 : 0));  // + the error word
      if (last) *err_appended = true;
      if (!dry) {
        const ReduceTreeArgs a{D, Y, Do, Yo, n_total, lv, first, last ? 1u : 0u, last ? err_src : nullptr};
        if (tt == 256) hipLaunchKernelGGL((k_reduce_tree<FID, 256>), dim3((n_total + 63) / 64), dim3(256), 0, c.stream, a);
        else hipLaunchKernelGGL((k_reduce_tree<FID, 512>), dim3((n_total + 127) / 128), dim3(512), 0, c.stream, a);
        HIPCHK(hipGetLastError());
      }
      D = Do, Y = Yo, n_in >>= lv, first = 0;
    }
    return Y;
  }
  void sort_pairs(uint32_t* k_in, uint32_t* k_out, uint32_t* v_in, uint32_t* v_out, size_t total,
                  uint32_t bits) {
    size_t tmp_bytes = 0;
    device_sort_pairs(nullptr, tmp_bytes, k_in, k_out, v_in, v_out, total, bits, c.stream);
    char* tmp = alloc<char>(tmp_bytes ? tmp_bytes : 1);
    if (dry) return;
    device_sort_pairs(tmp, tmp_bytes, k_in, k_out, v_in, v_out, total, bits, c.stream);
  }
  // Small results (window sums, error word) land in the context's pinned buffer and are copied to `dst` after the
  // stream sync: a pageable destination would make hipMemcpyAsync stage the copy synchronously.
  static constexpr size_t kPinnedBytes = 64 << 10;
  struct Landing {
    void* dst;
    size_t off, bytes;
  };
  std::vector<Landing> landings;
  size_t pinned_used = 0;
  void d2h(void* dst, const void* src, size_t bytes) {
    if (dry) return;
    if (!c.pinned) HIPCHK(hipHostMalloc((void**)&c.pinned, kPinnedBytes, hipHostMallocDefault));
    if (pinned_used + bytes <= kPinnedBytes) {
      HIPCHK(hipMemcpyAsync(c.pinned + pinned_used, src, bytes, hipMemcpyDeviceToHost, c.stream));
      landings.push_back({dst, pinned_used, bytes});
      pinned_used += (bytes + 63) & ~(size_t)63;
    } else {
      HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c.stream));
    }
  }
  // one copy of bytes1 + bytes2 contiguous device bytes, landing in two host destinations
  void d2h_split(void* dst1, size_t bytes1, void* dst2, size_t bytes2, const void* src) {
    if (dry) return;
    if (!c.pinned) HIPCHK(hipHostMalloc((void**)&c.pinned, kPinnedBytes, hipHostMallocDefault));
    if (pinned_used + bytes1 + bytes2 <= kPinnedBytes) {
      HIPCHK(hipMemcpyAsync(c.pinned + pinned_used, src, bytes1 + bytes2, hipMemcpyDeviceToHost, c.stream));
      landings.push_back({dst1, pinned_used, bytes1});
      landings.push_back({dst2, pinned_used + bytes1, bytes2});
      pinned_used += (bytes1 + bytes2 + 63) & ~(size_t)63;
    } else {
      d2h(dst1, src, bytes1);
      d2h(dst2, (const char*)src + bytes1, bytes2);
    }
  }
  void sync() {
    if (dry) return;
    stream_wait(c.stream);
    for (const Landing& l : landings) memcpy(l.dst, c.pinned + l.off, l.bytes);
    landings.clear();
    pinned_used = 0;
  }
  void mark(const char*) {
    if (dry || !prof) return;
    if (!c.have_ev) {
      for (int i = 0; i < kMaxMarks; i++) HIPCHK(hipEventCreate(&c.ev[i]));
      c.have_ev = true;
    }
    if (nmarks < kMaxMarks) HIPCHK(hipEventRecord(c.ev[nmarks++], c.stream));
  }
};

// ---------------------------------------------------------------------------------------------------
// one MSM request, and the per-curve operation table (each curve is instantiated in its own TU)
// ---------------------------------------------------------------------------------------------------
struct MsmCall {
  const void* scalars;  // host or device
  bool scalars_device;
  bool scalars_mont;
  uint32_t u64_bits;  // 0 => field scalars; NMX_BITS_AUTO resolved by the caller
  bool u64_mode;
  // precomputed tables of the registered key (0 = none / not used for this call)
  uint32_t pre_stride = 0, pre_offset = 0, pre_c = 0;
  // sparse forms: host base indices (validated < key length by the caller), and "all scalars are 1"
  const uint32_t* gather_host = nullptr;
  bool all_ones = false;
  bool bases_clean = false;  // the key holds no identity point
  // NMX_SCALARS_SHARDED: `scalars` is a host array of device pointers, one per piece of the call (nmx_shard_plan order), each
  // on the device that holds that piece of the key (resolved by key_msm before the per-curve code sees the call)
  bool scalars_sharded = false;
};

// one vector of a fused batch (msm_key_batch): n field scalars, host or device as the shared MsmCall says
struct BatchItem {
  const void* scalars;
  size_t n;
};

// Fills the freshly allocated device key (n x 64 raw bytes) -- nullptr: one hipMemcpy from `src`; key files stream
// through pinned staging buffers (keyfile.hip)
using BaseFill = std::function<void(void* d_dst, hipStream_t stream)>;

struct CurveOps {
  // out = sum scalars[i] * key[offset + i], i < n, through the key's window tables when it has them
  void (*msm_key)(Ctx&, const BaseSet&, size_t offset, size_t n, const MsmCall&, uint32_t flags, uint8_t* out,
                  uint8_t* inf);
  // Fused batch (a7, traits.rs:82-90 / hyperkzg.rs:593-612): out[j] = sum items[j].scalars[i] * key[offset + i], all k
  // vectors in ONE pipeline run over the key's tables (one bucket set per vector).  batch_limit: how many vectors one
  // run can take for this key (0: the key has no tables or its window width leaves no key bits for vector ids).
  void (*msm_key_batch)(Ctx&, const BaseSet&, size_t offset, const BatchItem* items, size_t k, const MsmCall& shared,
                        uint32_t flags, uint8_t* out64, uint8_t* inf);
  uint32_t (*batch_limit)(const BaseSet&);
  // msm_key(v) + h * r
  void (*commit)(Ctx&, const BaseSet&, size_t n, const MsmCall&, const void* h_xy64, const void* r, uint32_t flags,
                 uint8_t* out, uint8_t* inf);
  // fill bs.d / pre_c / pre_W / any_identity for bs.curve, bs.n: from a host / device array, or through `fill`
  void (*upload)(Ctx&, BaseSet& bs, const void* src, uint32_t flags, const BaseFill* fill);
  // host: one point in the ABI form (flags & NMX_BASES_MONT) -> canonical x||y; false if not canonical / off the curve
  bool (*check_point_host)(const uint8_t* xy64, uint32_t flags, uint8_t* out_canonical_xy64);
  const uint32_t* base_modulus_words;  // 8 x u32
  void (*generate)(Ctx&, BaseSet& bs, uint64_t k0, uint32_t flags);
  void (*internal_to_canonical)(uint8_t* elems32, size_t count);  // host, in place
  // host: sum of 128-byte partials -> affine point, or (flags & NMX_OUT_PARTIAL) one more partial
  void (*point_sum)(const uint8_t* partials128, size_t count, uint32_t flags, uint8_t* out, uint8_t* inf);
  // host: the blinding term h * r of `commit` as a 128-byte partial (h, r in the ABI forms `flags` names)
  void (*blind_term)(const void* h_xy64, const void* r, uint32_t flags, uint8_t* out128);
  bool (*check_layout)(const uint8_t* generator_raw64, const uint8_t* scalar_raw32, uint64_t value);  // host
  size_t (*table_bytes)(size_t n);  // host: HBM bytes of an n-point key with its window tables
  // out[j] = msm(items[j]) + h * rs[j] for k vectors over one key: ONE fused pipeline run when the key's tables take k bucket sets
  // (batch_limit), one run per vector otherwise; the k blinding terms on pool threads under the device work (as `commit`).
  // rs32: k scalars in the ABI form of `flags`, host.  The two commitments of an inner-product-argument round (ipa_pc.rs:213-232).
  // late: when set, the scalars are not known yet -- (*late)(j, out32) blocks until scalar j is and writes it (called on the pool
  // thread that computes blinding term j, under the device work); rs32 is ignored then.
  void (*commit_batch)(Ctx&, const BaseSet&, const BatchItem* items, size_t k, const MsmCall& shared, const void* h_xy64,
                       const uint8_t* rs32, const std::function<void(size_t, uint8_t*)>* late, uint32_t flags, uint8_t* out64, uint8_t* inf);
  int scalar_field;  // field id (NMX_F_*) of the curve's scalars
};
// field-vector kernels (fieldvec.hip)
void fv_axpy(Ctx&, int field, const void* a, const void* b, const void* r, size_t n, uint32_t flags, void* out);
void fv_axpy2(Ctx&, int field, const void* a, const void* b, const void* c, const void* r, size_t n, uint32_t flags,
              void* out);
void fv_cross_term(Ctx&, int field, const void* az, const void* bz, const void* cz, const void* e, const void* u,
                   size_t n, uint32_t flags, void* out);
void fv_cross_term2(Ctx&, int field, const void* az, const void* bz, const void* cz, const void* e1, const void* e2,
                    const void* u, size_t n, uint32_t flags, void* out);
void fv_vec_add(Ctx&, int field, const void* a, const void* b, size_t n, uint32_t flags, void* out);
void fv_bind(Ctx&, int field, const void* z, size_t z_len, size_t lo_off, size_t hi_off, size_t stride, const void* r,
             size_t n_out, uint32_t flags, void* out);

void fv_suffix_horner(Ctx&, int field, const void* f, size_t n, const void* u, uint32_t flags, void* out);
void fv_r1cs_cross_term(Ctx&, int field, const uint32_t* const* indptr, const uint32_t* const* indices, const uint32_t* const* data,
                        size_t rows, size_t cols, const void* z1, const void* z2, const void* e, const void* u, uint32_t flags, void* out);
void fv_nifs_fold(Ctx&, int field, const void* w1, const void* w2, size_t n_w, const void* e1, const void* t, size_t n_e, const void* r,
                  uint32_t flags, void* w, void* e);
void fv_eq_evals(Ctx&, int field, const void* r_host, uint32_t ell, uint32_t flags, uint32_t* d_out);
void fv_eq_evals_pair(Ctx&, int field, const void* r_host, uint32_t ellL, uint32_t ellR, uint32_t flags, uint32_t* d_outL,
                      uint32_t* d_outR);  // both tables of an evaluation, one launch when both fit the direct kernel
void fv_spmv_convert(Ctx&, int field, uint32_t* d_data, size_t nnz, uint32_t flags);
void fv_spmv_classify(Ctx&, int field, const uint32_t* d_data, uint32_t* d_indices, size_t nnz, size_t cols);
void fv_spmv_apply(Ctx&, int field, const uint32_t* indptr, const uint32_t* indices, const uint32_t* data, size_t rows,
                   size_t cols, const void* z, uint32_t flags, void* out);
void fv_spmv_apply_pair(Ctx&, int field, const uint32_t* indptr, const uint32_t* indices, const uint32_t* data,
                        size_t rows, size_t cols, const void* z1, const void* z2, uint32_t flags, void* out1, void* out2);
bool fv_batch_invert(Ctx&, int field, const void* v, size_t n, uint32_t flags, void* out);  // false: an element is zero
void fv_lincomb(Ctx&, int field, const void* const* vecs, const size_t* lens, size_t k, const void* s, size_t n_out,
                uint32_t flags, void* out);
// inner-product argument, field side (ipa.hpp): device words; `done` is recorded behind the round's kernel, whose block partials land in
// pinned host memory at *partial_host (the context's mailbox) -- fv_ipa_scalar adds them up on whatever thread waited for `done`
void fv_ipa_round(Ctx&, int field, const uint32_t* a, const uint32_t* b, const uint32_t* S, uint32_t* a_out, uint32_t* b_out, uint32_t* S_out,
                  size_t n, size_t len, const void* r_prev, const void* rinv_prev, uint32_t flags, uint32_t* vL, uint32_t* vR, hipEvent_t done,
                  const uint32_t** partial_host);
void fv_ipa_scalar(int field, const uint32_t* partial_host, size_t n, int which, uint32_t flags, uint8_t* out32);
void fv_ipa_last(Ctx&, int field, const uint32_t* a, const void* r, const void* rinv, uint32_t flags, uint32_t* dout, uint8_t* out32);
void fv_ipa_one(Ctx&, int field, uint32_t* S);
bool fv_ipa_invert(int field, const void* r, uint32_t flags, void* out);
void fv_plain_sums(Ctx&, int field, int kind, const void* A, const void* B, const void* C, size_t len, uint32_t flags,
                   uint8_t* out);  // sumcheck.hip
void fv_eval_multi(Ctx&, int field, const void* const* polys, const size_t* lens, size_t k, const void* points, size_t m,
                   uint32_t flags, uint8_t* out);  // sumcheck.hip
void fv_bind_eq_sums(Ctx&, int field, int mode, const void* A, const void* B, const void* C, size_t len, const void* r,
                     const void* eqL, size_t nL, const void* eqR, size_t nR, uint32_t shift, uint32_t flags, void* oA,
                     void* oB, void* oC, uint8_t* out);  // sumcheck.hip
void fv_fold_chain(Ctx&, int field, const void* p, size_t len, const void* xs, size_t k, uint32_t flags, void* const* outs);  // fieldvec.hip
struct SpmvManyItem {  // one matrix of nmx_spmv_apply_many (fieldvec.hip spmv_many_t)
  const uint32_t *indptr = nullptr, *indices = nullptr, *data = nullptr;                                                  // CSR (forward)
  const uint32_t *vptr = nullptr, *tix = nullptr, *tdata = nullptr, *vout = nullptr, *hrow = nullptr, *hstart = nullptr;  // M^T in virtual rows
  size_t nvirt = 0, nheavy = 0, nparts = 0, rows = 0, cols = 0;
  void* out = nullptr;
};
void fv_spmv_many(Ctx&, int field, const SpmvManyItem* items, size_t k, bool transposed, const void* x, uint32_t flags);
void fv_mle_multi_eval(Ctx&, int field, const void* const* zs, size_t k, size_t len, const uint32_t* eqL, const uint32_t* eqR, uint32_t s_right,
                       uint32_t flags, uint8_t* out);  // sumcheck_prove.hpp: HBM-resident polynomials, results through the mailbox
void fv_eq_sums(Ctx&, int field, int mode, const void* A, const void* B, const void* C, size_t len, const void* eqL,
                size_t nL, const void* eqR, size_t nR, uint32_t shift, uint32_t flags, uint8_t* out);  // sumcheck.hip
void fv_spmv_apply_transposed(Ctx&, int field, const uint32_t* vptr, const uint32_t* indices, const uint32_t* data, const uint32_t* vout,
                              const uint32_t* hrow, const uint32_t* hstart, size_t nvirt, size_t nheavy, size_t nparts, size_t rows, size_t cols,
                              const void* x, uint32_t flags, void* out);
// Spartan's sum-check provers, one call each (sumcheck_prove.hpp); which = 3: cubic with three inputs, 4: quad_prod
using TranscriptFn = int (*)(void* ctx, const uint8_t* coeffs, size_t n_coeffs, uint8_t* challenge32);
void fv_sumcheck_prove(Ctx&, int field, int which, const void* claim, const void* taus, size_t num_rounds, void* A, void* B, void* C,
                       uint32_t flags, TranscriptFn cb, void* cb_ctx, uint8_t* out_polys, uint8_t* out_r, uint8_t* out_claims);
void fv_sumcheck_prove_batch(Ctx&, int field, const uint8_t* claims, const size_t* num_rounds, void* const* polys,
                             const uint8_t* const* eq_points, const uint8_t* coeffs, size_t k, uint32_t flags, TranscriptFn cb, void* cb_ctx,
                             uint8_t* out_polys, uint8_t* out_r, uint8_t* out_finals);

const CurveOps& curve_ops_bn254_g1();
const CurveOps& curve_ops_grumpkin();
const CurveOps& curve_ops_pallas();
const CurveOps& curve_ops_vesta();

}  // namespace nmx
