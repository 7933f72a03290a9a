// bench/graph_gap.hip -- what would a hipGraph buy the MSM's dependent launch chain?  (VERDICT r2 next #1a)
//
// One MSM is ~25 dependent launches on one stream (partition 5, accumulate 1, fold 3, reduction tree 4-16, copies), most of
// them a few microseconds long.  This measures, for a chain of N dependent kernels of the same kind (one small block that
// bumps a counter, or a 256-block grid that does ~5 us of work):
//   * wall time per chain when launched kernel by kernel on a stream  (what the library does)
//   * wall time per chain when the same chain is captured once and replayed with hipGraphLaunch
//   * host time spent enqueueing in both cases
// Build: hipcc -O2 --offload-arch=gfx950 -o bench/graph_gap bench/graph_gap.hip ; run: bench/graph_gap [chain_len] [reps]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ void k_tiny(uint32_t* p) {
  if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1;
}
// ~`iters` dependent multiply-adds per lane: 256 blocks x 256 lanes, a few microseconds
__global__ void k_work(uint32_t* p, uint32_t iters) {
  uint32_t v = p[(blockIdx.x * 256u + threadIdx.x) & 1023u];
  for (uint32_t i = 0; i < iters; i++) v = v * 2654435761u + i;
  if (v == 0x12345u) p[1024] = v;
}
static double now_us() {
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
struct Res { double wall_us, host_us; };
template <class Enq> Res timed(hipStream_t s, int reps, Enq enq) {
  std::vector<double> wall, host;
  for (int r = 0; r < reps + 3; r++) {
    CHK(hipStreamSynchronize(s));
    double t0 = now_us();
    enq();
    double t1 = now_us();
    CHK(hipStreamSynchronize(s));
    double t2 = now_us();
    if (r >= 3) { wall.push_back(t2 - t0); host.push_back(t1 - t0); }
  }
  std::sort(wall.begin(), wall.end()); std::sort(host.begin(), host.end());
  return {wall[wall.size() / 2], host[host.size() / 2]};
}
int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 24, reps = argc > 2 ? atoi(argv[2]) : 200;
  hipStream_t s; CHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  uint32_t* d; CHK(hipMalloc(&d, 8192)); CHK(hipMemset(d, 0, 8192));
  for (int kind = 0; kind < 2; kind++) {
    const uint32_t iters = 2000;
    auto chain = [&]() {
      for (int i = 0; i < N; i++) {
        if (kind == 0) k_tiny<<<1, 64, 0, s>>>(d);
        else k_work<<<256, 256, 0, s>>>(d, iters);
      }
    };
    // single kernel alone: its own duration
    Res one = timed(s, reps, [&]() { if (kind == 0) k_tiny<<<1, 64, 0, s>>>(d); else k_work<<<256, 256, 0, s>>>(d, iters); });
    Res st = timed(s, reps, chain);
    hipGraph_t g; hipGraphExec_t ge;
    CHK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    chain();
    CHK(hipStreamEndCapture(s, &g));
    CHK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    Res gr = timed(s, reps, [&]() { CHK(hipGraphLaunch(ge, s)); });
    printf("{\"kernel\": \"%s\", \"chain\": %d, \"one_launch_wall_us\": %.2f, \"stream_wall_us\": %.2f, \"stream_host_us\": %.2f, "
           "\"graph_wall_us\": %.2f, \"graph_host_us\": %.2f, \"stream_per_link_us\": %.3f, \"graph_per_link_us\": %.3f}\n",
           kind == 0 ? "tiny (1 block)" : "work (256 blocks)", N, one.wall_us, st.wall_us, st.host_us, gr.wall_us, gr.host_us,
           (st.wall_us - one.wall_us) / (N - 1), (gr.wall_us - one.wall_us) / (N - 1));
    CHK(hipGraphExecDestroy(ge)); CHK(hipGraphDestroy(g));
  }
  return 0;
}
