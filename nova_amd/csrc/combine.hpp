// combine.hpp -- the exchange step of a sharded MSM inside ONE host process: one 128-byte partial per GPU, all-gathered over
// RCCL (xGMI) and summed.  SURVEY.md 8(e): RCCL has no elliptic-curve reduce op, so north_star's "final RCCL reduce of partial
// bucket sums" is an ncclAllGather of raw bytes followed by a G-term point sum -- the reference's rayon
// `reduce(identity, +)`, /root/reference/src/provider/msm.rs:566-571,667-673.  Included by capi.hip only.
//
// RCCL is bound at run time (dlopen): a Rust host that never shards a key does not load it, a Python host that imported
// torch shares torch's copy (same soname), and a box without RCCL still works -- the combine then is the host sum of the
// partials the shard workers already hold, which `nmx_profile_last_sharded` reports as rccl_ranks = 0.
//
// One communicator per PHYSICAL device in use (ncclCommInitAll; logical devices that share a GPU under
// NMX_DEVICES_OVERSUBSCRIBE are summed on the host first: RCCL refuses two ranks on one device).  A collective is issued by
// ONE thread for all ranks inside ncclGroupStart / ncclGroupEnd -- the single-process multi-GPU pattern -- and calls from
// different host threads (the trait is called from rayon workers) are serialised: every rank must see the collectives of
// a communicator in the same order.
#pragma once
#include <dlfcn.h>

#include "runtime.hpp"

namespace nmx {

struct RcclApi {
  using Comm = void*;
  int (*CommInitAll)(Comm*, int, const int*) = nullptr;
  int (*CommDestroy)(Comm) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int /* ncclDataType_t */, Comm, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  void* lib = nullptr;
  bool load() {
    if (lib) return true;
    // the copy already in the process first (torch's), then the ROCm one through this library's RUNPATH
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) return false;
    auto sym = [&](const char* n) { return dlsym(h, n); };
    CommInitAll = (decltype(CommInitAll))sym("ncclCommInitAll");
    CommDestroy = (decltype(CommDestroy))sym("ncclCommDestroy");
    AllGather = (decltype(AllGather))sym("ncclAllGather");
    GroupStart = (decltype(GroupStart))sym("ncclGroupStart");
    GroupEnd = (decltype(GroupEnd))sym("ncclGroupEnd");
    GetErrorString = (decltype(GetErrorString))sym("ncclGetErrorString");
    if (!CommInitAll || !CommDestroy || !AllGather || !GroupStart || !GroupEnd) return false;
    lib = h;
    return true;
  }
};

// payload of one rank: the partial (XYZZW, 128 bytes) -- padded to 16-byte multiples so that every rank's slot is aligned
static constexpr size_t kCombineSlot = 128;

struct RcclCombine {
  std::mutex mu;  // one collective at a time (see the header comment)
  RcclApi api;
  std::vector<int> hip_devs;  // rank r runs on HIP device hip_devs[r]
  std::vector<RcclApi::Comm> comms;
  std::vector<hipStream_t> streams;
  std::vector<uint8_t*> send, recv;  // device buffers per rank: kCombineSlot and kCombineSlot * ranks bytes
  uint8_t* pinned = nullptr;         // host staging: ranks slots out, ranks slots back
  bool failed = false;               // set once: RCCL missing or a communicator could not be created -> host sums from then on
  std::string why;

  int ranks() const { return (int)comms.size(); }
  void destroy() {  // G.mu not needed: called from nmx_shutdown / a device-set change with no call in flight
    std::lock_guard<std::mutex> lk(mu);
    destroy_locked();
    failed = false;
    why.clear();
  }
  void destroy_locked() {
    int prev = -1;
    (void)hipGetDevice(&prev);
    for (size_t r = 0; r < comms.size(); r++) {
      (void)hipSetDevice(hip_devs[r]);
      if (streams[r]) (void)hipStreamSynchronize(streams[r]);
      if (comms[r]) (void)api.CommDestroy(comms[r]);
      if (send[r]) (void)hipFree(send[r]);
      if (recv[r]) (void)hipFree(recv[r]);
      if (streams[r]) (void)hipStreamDestroy(streams[r]);
    }
    if (pinned) (void)hipHostFree(pinned);
    pinned = nullptr;
    comms.clear(), streams.clear(), send.clear(), recv.clear(), hip_devs.clear();
    if (prev >= 0) (void)hipSetDevice(prev);
  }
  // communicator over exactly `devs` (distinct HIP devices, in rank order); false: not available (reason in `why`)
  bool ensure_locked(const std::vector<int>& devs) {
    if (failed) return false;
    if (devs == hip_devs && !comms.empty()) return true;
    destroy_locked();
    if (!api.load()) {
      failed = true;
      why = "librccl.so.1 not found";
      return false;
    }
    const int n = (int)devs.size();
    std::vector<RcclApi::Comm> cs((size_t)n, nullptr);
    const int rc = api.CommInitAll(cs.data(), n, devs.data());
    if (rc != 0) {
      failed = true;
      why = std::string("ncclCommInitAll: ") + (api.GetErrorString ? api.GetErrorString(rc) : "error");
      (void)hipGetLastError();
      return false;
    }
    hip_devs = devs;
    comms = cs;
    streams.assign((size_t)n, nullptr);
    send.assign((size_t)n, nullptr);
    recv.assign((size_t)n, nullptr);
    try {
      for (int r = 0; r < n; r++) {
        HIPCHK(hipSetDevice(devs[(size_t)r]));
        HIPCHK(hipStreamCreateWithFlags(&streams[(size_t)r], hipStreamNonBlocking));
        HIPCHK(hipMalloc((void**)&send[(size_t)r], kCombineSlot));
        HIPCHK(hipMalloc((void**)&recv[(size_t)r], kCombineSlot * (size_t)n));
      }
      HIPCHK(hipHostMalloc((void**)&pinned, 2 * kCombineSlot * (size_t)n, hipHostMallocDefault));
    } catch (const Fail& f) {
      destroy_locked();
      failed = true;
      why = f.msg;
      return false;
    }
    return true;
  }
  // all-gather of one slot per rank; gathered[ranks * kCombineSlot] <- every rank's slot, read back from rank 0.
  // Throws Fail on a HIP / RCCL error in flight (the caller falls back to the host sum of what it holds).  An error never
  // leaves this thread inside an open ncclGroupStart (later collectives of the thread -- torch's too, the dlopen'ed librccl
  // is shared -- would queue into it and hang): the group is closed by a guard before the exception travels, and the
  // communicators are torn down and `failed` set, so that every later call of the process takes the host sum.
  void all_gather_locked(const uint8_t* slots /* ranks * kCombineSlot, host */, uint8_t* gathered) {
    try {
      all_gather_impl(slots, gathered);
    } catch (const Fail& f) {
      (void)hipGetLastError();
      destroy_locked();
      failed = true;
      why = "RCCL combine failed in flight: " + f.msg;
      throw;
    }
  }

 private:
  void all_gather_impl(const uint8_t* slots, uint8_t* gathered) {
    const size_t n = comms.size();
    memcpy(pinned, slots, n * kCombineSlot);
    for (size_t r = 0; r < n; r++) {
      HIPCHK(hipSetDevice(hip_devs[r]));
      HIPCHK(hipMemcpyAsync(send[r], pinned + r * kCombineSlot, kCombineSlot, hipMemcpyHostToDevice, streams[r]));
    }
    auto chk = [&](int rc, const char* what) {
      if (rc != 0) throw Fail{NMX_E_HIP, std::string(what) + ": " + (api.GetErrorString ? api.GetErrorString(rc) : "RCCL error")};
    };
    chk(api.GroupStart(), "ncclGroupStart");
    {
      struct GroupGuard {  // ncclGroupEnd on every path out of the bracket
        RcclApi& api;
        bool open = true;
        int end() {
          open = false;
          return api.GroupEnd();
        }
        ~GroupGuard() {
          if (open) (void)api.GroupEnd();
        }
      } guard{api};
      for (size_t r = 0; r < n; r++) {
        HIPCHK(hipSetDevice(hip_devs[r]));
        chk(api.AllGather(send[r], recv[r], kCombineSlot, 1 /* ncclUint8 */, comms[r], streams[r]), "ncclAllGather");
      }
      chk(guard.end(), "ncclGroupEnd");
    }
    HIPCHK(hipSetDevice(hip_devs[0]));
    HIPCHK(hipMemcpyAsync(pinned + n * kCombineSlot, recv[0], n * kCombineSlot, hipMemcpyDeviceToHost, streams[0]));
    for (size_t r = 0; r < n; r++) {  // every rank's collective has completed before the buffers are reused
      HIPCHK(hipSetDevice(hip_devs[r]));
      HIPCHK(hipStreamSynchronize(streams[r]));
    }
    memcpy(gathered, pinned + n * kCombineSlot, n * kCombineSlot);
  }
};

}  // namespace nmx
