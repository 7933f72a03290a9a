// msm.hip -- libnova_mi355x.so: device backend + the C ABI of include/nova_mi355x.h.
//
// Replaces, behind the reference's DlogGroupExt / CommitmentEngineTrait seam:
//   /root/reference/src/provider/msm.rs:225-419,469-503  (msm, msm_small, msm_small_with_max_num_bits)
//   halo2curves::msm::msm_best (called at msm.rs:411,500)
//   /root/reference/src/provider/traits.rs:82-90         (batch_vartime_multiscalar_mul default)
//   /root/reference/src/provider/pedersen.rs:263-270, hyperkzg.rs:584-591 (commit = msm + h*r)
// There is no CPU fallback in this file: without a HIP device every entry point returns NMX_E_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <cstring>

#include <rocprim/rocprim.hpp>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/nova_mi355x.h"
#include "curves.hpp"
#include "msm_pipeline.hpp"

namespace nmx {

// ---------------------------------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------------------------------
static thread_local std::string t_err;
struct Fail {
  int code;
  std::string msg;
};
#define HIPCHK(x)                                                                                  \
  do {                                                                                             \
    hipError_t e_ = (x);                                                                           \
    if (e_ != hipSuccess)                                                                          \
      throw Fail{NMX_E_HIP, std::string(#x) + ": " + hipGetErrorString(e_)};                       \
  } while (0)
static inline void require(bool ok, int code, const char* msg) {
  if (!ok) throw Fail{code, msg};
}

// ---------------------------------------------------------------------------------------------------
// generic launch trampoline: one lane per tid
// ---------------------------------------------------------------------------------------------------
template <class F> __global__ __launch_bounds__(256) void k_launch(F f, uint32_t n) {
  uint32_t tid = blockIdx.x * 256u + threadIdx.x;
  if (tid < n) f(tid);
}

// ---------------------------------------------------------------------------------------------------
// per-call context: stream + workspace arena + profiling events
// ---------------------------------------------------------------------------------------------------
static constexpr int kMaxMarks = 12;
struct Ctx {
  hipStream_t stream = nullptr;
  char* arena = nullptr;
  size_t cap = 0;
  hipEvent_t ev[kMaxMarks];
  bool have_ev = false;
};

struct Global {
  std::mutex mu;
  bool inited = false;
  int device = 0;
  std::vector<Ctx*> free_ctx;
  std::vector<Ctx*> all_ctx;
  struct BaseSet {
    int curve;
    size_t n;
    void* d;  // Affine<BF>[n], Montgomery
  };
  std::unordered_map<uint64_t, BaseSet> bases;
  uint64_t next_handle = 1;
  bool profiling = false;
  uint32_t force_c = 0;
};
static Global G;
static thread_local float t_prof[kMaxMarks];
static thread_local int t_prof_n = 0;

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
  HIPCHK(hipSetDevice(dev));
  G.inited = true;
}

struct CtxLease {
  Ctx* c;
  CtxLease() {
    ensure_init();
    HIPCHK(hipSetDevice(G.device));
    {
      std::lock_guard<std::mutex> lk(G.mu);
      if (!G.free_ctx.empty()) {
        c = G.free_ctx.back();
        G.free_ctx.pop_back();
        return;
      }
    }
    c = new Ctx();
    HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    std::lock_guard<std::mutex> lk(G.mu);
    G.all_ctx.push_back(c);
  }
  ~CtxLease() {
    std::lock_guard<std::mutex> lk(G.mu);
    G.free_ctx.push_back(c);
  }
};

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
  void sort_pairs(uint32_t* k_in, uint32_t* k_out, uint32_t* v_in, uint32_t* v_out, size_t total,
                  uint32_t bits) {
    size_t tmp_bytes = 0;
    HIPCHK(rocprim::radix_sort_pairs(nullptr, tmp_bytes, k_in, k_out, v_in, v_out, total, 0u, bits,
                                     c.stream));
    char* tmp = alloc<char>(tmp_bytes ? tmp_bytes : 1);
    if (dry) return;
    HIPCHK(rocprim::radix_sort_pairs(tmp, tmp_bytes, k_in, k_out, v_in, v_out, total, 0u, bits, c.stream));
  }
  void d2h(void* dst, const void* src, size_t bytes) {
    if (dry) return;
    HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c.stream));
  }
  void sync() {
    if (dry) return;
    HIPCHK(hipStreamSynchronize(c.stream));
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

static void arena_reserve(Ctx& c, size_t bytes) {
  if (bytes <= c.cap) return;
  if (c.arena) HIPCHK(hipFree(c.arena));
  c.arena = nullptr;
  c.cap = 0;
  size_t want = bytes + bytes / 8 + (1u << 20);
  HIPCHK(hipMalloc((void**)&c.arena, want));
  c.cap = want;
}

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
struct OrFn {  // OR of all u64 scalars -> bit length of the maximum
  const uint32_t* s;
  uint32_t* out;
  NMX_HD void operator()(uint32_t i) const {
    if (s[2 * (size_t)i]) nmx_atomic_or(out, s[2 * (size_t)i]);
    if (s[2 * (size_t)i + 1]) nmx_atomic_or(out + 1, s[2 * (size_t)i + 1]);
  }
};
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
struct MsmCall {
  const void* scalars;   // host or device
  bool scalars_device;
  bool scalars_mont;
  uint32_t u64_bits;     // 0 => field scalars; NMX_BITS_AUTO resolved by the caller
  bool u64_mode;
};

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
  a.force_c = G.force_c;
  {
    uint32_t bits = a.u64_bits ? a.u64_bits : sbits;
    MsmShape sh = make_shape(a.n, bits, a.force_c);
    require((uint64_t)n * sh.W < 0xffffffffull && n < 0x7fffffffull, NMX_E_TOO_LARGE,
            "n * windows must be < 2^32");
  }
  XYZZW wsum[260];
  uint32_t err = 0;
  MsmShape sh{};
  const bool prof = G.profiling;
  for (int pass = 0; pass < 2; pass++) {
    DeviceBackend be(c, pass == 0, prof);
    if (mc.scalars_device) {
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
    } else if (prof) {
      t_prof_n = 0;
      for (int i = 0; i + 1 < be.nmarks; i++) {
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, c.ev[i], c.ev[i + 1]));
        t_prof[t_prof_n++] = ms;
      }
    }
  }
  require(!(err & ERR_SCALAR_RANGE), NMX_E_SCALAR_RANGE, "scalar >= field modulus");
  require(!(err & ERR_SMALL_RANGE), NMX_E_SMALL_RANGE, "small scalar >= 2^max_num_bits");
  auto t0 = std::chrono::steady_clock::now();
  XYZZ<BF> r = combine_windows<BF>(wsum, sh);
  if (prof && t_prof_n > 0) {
    auto t1 = std::chrono::steady_clock::now();
    t_prof[t_prof_n - 1] += std::chrono::duration<float, std::milli>(t1 - t0).count();
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
template <int CID> static void* upload_bases(Ctx& c, const void* src, size_t n, uint32_t flags) {
  constexpr int BF = CurveT<CID>::BF;
  void* d = nullptr;
  if (n == 0) return nullptr;
  HIPCHK(hipMalloc(&d, n * 64));
  try {
    HIPCHK(hipMemcpyAsync(d, src, n * 64,
                          (flags & NMX_BASES_DEVICE) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice,
                          c.stream));
    {
      DeviceBackend be(c, false, false);
      ToInternalFn<BF> f{(uint32_t*)d, (flags & NMX_BASES_MONT) ? 1u : 0u};
      be.launch(f, (uint32_t)(2 * n));
    }
    HIPCHK(hipStreamSynchronize(c.stream));
  } catch (...) {
    (void)hipFree(d);
    throw;
  }
  return d;
}

static Global::BaseSet lookup(uint64_t h) {
  std::lock_guard<std::mutex> lk(G.mu);
  auto it = G.bases.find(h);
  if (it == G.bases.end()) throw Fail{NMX_E_HANDLE, "unknown base handle"};
  return it->second;
}

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

template <int CID>
static void msm_entry(const void* d_bases, size_t n, const MsmCall& mc, uint32_t flags, uint8_t* out,
                      uint8_t* is_inf, Ctx& c) {
  auto r = run_msm<CID>(c, d_bases, n, mc);
  write_result<CID>(r, flags, out, is_inf);
}

#define DISPATCH_CURVE(curve, ...)                              \
  switch (curve) {                                              \
    case 0: { constexpr int CID = 0; __VA_ARGS__; } break;      \
    case 1: { constexpr int CID = 1; __VA_ARGS__; } break;      \
    case 2: { constexpr int CID = 2; __VA_ARGS__; } break;      \
    case 3: { constexpr int CID = 3; __VA_ARGS__; } break;      \
    default: throw Fail{NMX_E_ARG, "bad curve id"};             \
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

struct TempBases {  // RAII for one-shot uploads
  void* d = nullptr;
  ~TempBases() {
    if (d) (void)hipFree(d);
  }
};

}  // namespace nmx

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

int nmx_shutdown(void) {
  return guarded([&] {
    std::lock_guard<std::mutex> lk(G.mu);
    if (!G.inited) return;
    (void)hipSetDevice(G.device);
    for (auto& kv : G.bases)
      if (kv.second.d) (void)hipFree(kv.second.d);
    G.bases.clear();
    for (Ctx* c : G.all_ctx) {
      if (c->arena) (void)hipFree(c->arena);
      if (c->have_ev)
        for (int i = 0; i < kMaxMarks; i++) (void)hipEventDestroy(c->ev[i]);
      if (c->stream) (void)hipStreamDestroy(c->stream);
      delete c;
    }
    G.all_ctx.clear();
    G.free_ctx.clear();
    G.inited = false;
  });
}

int nmx_device_count(void) {
  int cnt = 0;
  if (hipGetDeviceCount(&cnt) != hipSuccess) return 0;
  return cnt;
}

const char* nmx_last_error(void) { return t_err.c_str(); }
const char* nmx_version(void) { return "nova-mi355x 0.1.0 (gfx950)"; }

int nmx_bases_register(int curve, const void* bases, size_t n, uint32_t flags, uint64_t* handle) {
  return guarded([&] {
    require(handle && (bases || n == 0), NMX_E_ARG, "null argument");
    CtxLease L;
    void* d = nullptr;
    DISPATCH_CURVE(curve, d = upload_bases<CID>(*L.c, bases, n, flags));
    std::lock_guard<std::mutex> lk(G.mu);
    uint64_t h = G.next_handle++;
    G.bases[h] = Global::BaseSet{curve, n, d};
    *handle = h;
  });
}

int nmx_bases_unregister(uint64_t handle) {
  return guarded([&] {
    ensure_init();
    Global::BaseSet bs;
    {
      std::lock_guard<std::mutex> lk(G.mu);
      auto it = G.bases.find(handle);
      if (it == G.bases.end()) throw Fail{NMX_E_HANDLE, "unknown base handle"};
      bs = it->second;
      G.bases.erase(it);
    }
    HIPCHK(hipSetDevice(G.device));
    if (bs.d) HIPCHK(hipFree(bs.d));
  });
}

int nmx_bases_read(uint64_t handle, size_t offset, size_t n, void* out_xy64) {
  return guarded([&] {
    require(out_xy64 || n == 0, NMX_E_ARG, "null argument");
    auto bs = lookup(handle);
    require(offset + n <= bs.n, NMX_E_HANDLE, "offset + n beyond the registered key");
    CtxLease L;
    HIPCHK(hipMemcpyAsync(out_xy64, (const char*)bs.d + offset * 64, n * 64, hipMemcpyDeviceToHost,
                          L.c->stream));
    HIPCHK(hipStreamSynchronize(L.c->stream));
    uint8_t* o = (uint8_t*)out_xy64;
    DISPATCH_CURVE(bs.curve, {
      constexpr int BF = CurveT<CID>::BF;
      for (size_t i = 0; i < 2 * n; i++) {
        Fp<BF> f = fp_from_bytes<BF>(o + 32 * i);
        fp_to_bytes(f.to_canonical(), o + 32 * i);
      }
    });
  });
}

int nmx_bases_generate(int curve, uint64_t k0, size_t n, uint64_t* handle) {
  return guarded([&] {
    require(handle != nullptr, NMX_E_ARG, "null argument");
    require(n < (1ull << 31) && k0 < (1ull << 62), NMX_E_ARG, "k0 / n out of range");
    CtxLease L;
    void* d = nullptr;
    if (n) HIPCHK(hipMalloc(&d, n * 64));
    try {
      DeviceBackend be(*L.c, false, false);
      DISPATCH_CURVE(curve, {
        GenFn<CID> f{(AffineW*)d, k0};
        be.launch(f, (uint32_t)n);
      });
      HIPCHK(hipStreamSynchronize(L.c->stream));
    } catch (...) {
      if (d) (void)hipFree(d);
      throw;
    }
    std::lock_guard<std::mutex> lk(G.mu);
    uint64_t h = G.next_handle++;
    G.bases[h] = Global::BaseSet{curve, n, d};
    *handle = h;
  });
}

int nmx_msm_handle(uint64_t handle, size_t offset, const void* scalars, size_t n, uint32_t flags,
                   uint8_t* out, uint8_t* out_is_inf) {
  return guarded([&] {
    require(out && (scalars || n == 0), NMX_E_ARG, "null argument");
    auto bs = lookup(handle);
    require(offset + n <= bs.n, NMX_E_HANDLE, "offset + n beyond the registered key");
    CtxLease L;
    MsmCall mc{scalars, (flags & NMX_SCALARS_DEVICE) != 0, (flags & NMX_SCALARS_MONT) != 0, 0, false};
    DISPATCH_CURVE(bs.curve,
                   msm_entry<CID>((const char*)bs.d + offset * 64, n, mc, flags, out, out_is_inf, *L.c));
  });
}

int nmx_msm(int curve, const void* scalars, const void* bases, size_t n, uint32_t flags, uint8_t* out,
            uint8_t* out_is_inf) {
  return guarded([&] {
    require(out && ((scalars && bases) || n == 0), NMX_E_ARG, "null argument");
    CtxLease L;
    TempBases tb;
    MsmCall mc{scalars, (flags & NMX_SCALARS_DEVICE) != 0, (flags & NMX_SCALARS_MONT) != 0, 0, false};
    DISPATCH_CURVE(curve, {
      tb.d = upload_bases<CID>(*L.c, bases, n, flags);
      msm_entry<CID>(tb.d, n, mc, flags, out, out_is_inf, *L.c);
    });
  });
}

int nmx_msm_u64_handle(uint64_t handle, size_t offset, const uint64_t* scalars, size_t n,
                       uint32_t max_num_bits, uint32_t flags, uint8_t* out, uint8_t* out_is_inf) {
  return guarded([&] {
    require(out && (scalars || n == 0), NMX_E_ARG, "null argument");
    auto bs = lookup(handle);
    require(offset + n <= bs.n, NMX_E_HANDLE, "offset + n beyond the registered key");
    CtxLease L;
    bool dev = (flags & NMX_SCALARS_DEVICE) != 0;
    uint32_t bits = resolve_u64_bits(*L.c, scalars, n, dev, max_num_bits);
    MsmCall mc{scalars, dev, false, bits, true};
    DISPATCH_CURVE(bs.curve,
                   msm_entry<CID>((const char*)bs.d + offset * 64, n, mc, flags, out, out_is_inf, *L.c));
  });
}

int nmx_msm_u64(int curve, const uint64_t* scalars, const void* bases, size_t n, uint32_t max_num_bits,
                uint32_t flags, uint8_t* out, uint8_t* out_is_inf) {
  return guarded([&] {
    require(out && ((scalars && bases) || n == 0), NMX_E_ARG, "null argument");
    CtxLease L;
    TempBases tb;
    bool dev = (flags & NMX_SCALARS_DEVICE) != 0;
    uint32_t bits = resolve_u64_bits(*L.c, scalars, n, dev, max_num_bits);
    MsmCall mc{scalars, dev, false, bits, true};
    DISPATCH_CURVE(curve, {
      tb.d = upload_bases<CID>(*L.c, bases, n, flags);
      msm_entry<CID>(tb.d, n, mc, flags, out, out_is_inf, *L.c);
    });
  });
}

static void batch_impl(int curve, const void* d_bases, size_t n_bases, const void* const* vecs,
                       const size_t* lens, size_t k, uint32_t flags, uint8_t* out, uint8_t* out_is_inf,
                       Ctx& c) {
  require((vecs && lens && out) || k == 0, NMX_E_ARG, "null argument");
  require(!(flags & NMX_OUT_PARTIAL), NMX_E_ARG, "NMX_OUT_PARTIAL is not supported for batches");
  for (size_t j = 0; j < k; j++) {
    require(lens[j] <= n_bases, NMX_E_ARG, "vector longer than the base array");  // traits.rs:88 slices bases[..len]
    require(vecs[j] || lens[j] == 0, NMX_E_ARG, "null scalar vector");
    MsmCall mc{vecs[j], (flags & NMX_SCALARS_DEVICE) != 0, (flags & NMX_SCALARS_MONT) != 0, 0, false};
    DISPATCH_CURVE(curve, msm_entry<CID>(d_bases, lens[j], mc, flags, out + 64 * j,
                                         out_is_inf ? out_is_inf + j : nullptr, c));
  }
}

int nmx_msm_batch_handle(uint64_t handle, const void* const* scalar_vecs, const size_t* lens, size_t k,
                         uint32_t flags, uint8_t* out, uint8_t* out_is_inf) {
  return guarded([&] {
    auto bs = lookup(handle);
    CtxLease L;
    batch_impl(bs.curve, bs.d, bs.n, scalar_vecs, lens, k, flags, out, out_is_inf, *L.c);
  });
}

int nmx_msm_batch(int curve, const void* const* scalar_vecs, const size_t* lens, size_t k,
                  const void* bases, size_t n_bases, uint32_t flags, uint8_t* out, uint8_t* out_is_inf) {
  return guarded([&] {
    require(bases || n_bases == 0, NMX_E_ARG, "null argument");
    CtxLease L;
    TempBases tb;
    DISPATCH_CURVE(curve, tb.d = upload_bases<CID>(*L.c, bases, n_bases, flags));
    batch_impl(curve, tb.d, n_bases, scalar_vecs, lens, k, flags, out, out_is_inf, *L.c);
  });
}

int nmx_commit(uint64_t ck_handle, const void* v, size_t n, const void* h_xy64, const void* r,
               uint32_t flags, uint8_t* out, uint8_t* out_is_inf) {
  return guarded([&] {
    require(out && (v || n == 0) && h_xy64 && r, NMX_E_ARG, "null argument");
    auto bs = lookup(ck_handle);
    require(n <= bs.n, NMX_E_HANDLE, "ck shorter than v");  // assert!(ck.ck.len() >= v.len()), pedersen.rs:264
    CtxLease L;
    MsmCall mc{v, (flags & NMX_SCALARS_DEVICE) != 0, (flags & NMX_SCALARS_MONT) != 0, 0, false};
    DISPATCH_CURVE(bs.curve, {
      constexpr int BF = CurveT<CID>::BF, SF = CurveT<CID>::SF;
      auto acc = run_msm<CID>(*L.c, bs.d, n, mc);
      uint32_t rw[8];
      memcpy(rw, r, 32);
      require(Fp<SF>::words_lt_p(rw), NMX_E_SCALAR_RANGE, "blinding scalar >= field modulus");
      if (flags & NMX_SCALARS_MONT) Fp<SF>::from_words(rw).mont256_to_canonical().to_words(rw);
      uint32_t any = 0;
      for (int i = 0; i < 8; i++) any |= rw[i];
      if (any) {
        Affine<BF> h;
        h.x = fp_from_bytes<BF>((const uint8_t*)h_xy64);
        h.y = fp_from_bytes<BF>((const uint8_t*)h_xy64 + 32);
        if (!h.is_identity()) {
          const bool m = (flags & NMX_BASES_MONT) != 0;
          h.x = (m ? h.x.mont256_to_internal() : h.x.to_internal()).canon();
          h.y = (m ? h.y.mont256_to_internal() : h.y.to_internal()).canon();
        }
        acc.add(scalar_mul<BF>(XYZZ<BF>::from_affine(h), rw));
      }
      write_result<CID>(acc, flags, out, out_is_inf);
    });
  });
}

int nmx_point_sum(int curve, const uint8_t* partials128, size_t count, uint8_t* out, uint8_t* out_is_inf) {
  return guarded([&] {
    require(out && (partials128 || count == 0), NMX_E_ARG, "null argument");
    DISPATCH_CURVE(curve, {
      constexpr int BF = CurveT<CID>::BF;
      XYZZ<BF> acc = XYZZ<BF>::identity();
      for (size_t i = 0; i < count; i++) {
        XYZZW w;
        memcpy(w.w, partials128 + 128 * i, 128);
        acc.add(XYZZ<BF>::load(w));
      }
      xyzz_to_xy64<BF>(acc, out, out_is_inf);
    });
  });
}

int nmx_set_profiling(int on) {
  G.profiling = on != 0;
  return NMX_OK;
}
int nmx_profile_last(float* ms, int cap) {
  int n = t_prof_n < cap ? t_prof_n : cap;
  for (int i = 0; i < n; i++) ms[i] = t_prof[i];
  return t_prof_n;
}
int nmx_set_window_bits(uint32_t c) {
  if (c > 24) return NMX_E_ARG;
  G.force_c = c;
  return NMX_OK;
}

}  // extern "C"
