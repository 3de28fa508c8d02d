// Probe: do two chains of latency-bound kernels overlap (a) as two branches of ONE captured hipGraph, (b) as two linear
// graphs replayed on two streams, (c) as plain launches on two streams?  Each kernel is one wave that idles T microseconds.
//   hipcc --offload-arch=gfx950 -O2 -o graph_overlap.bin graph_overlap.hip && ./graph_overlap.bin [N per chain] [T us]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("{\"error\": \"%s at line %d\"}\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(64) void idle_kernel(long ticks) {          // exit: the 100 MHz clock passes the deadline
  const long t0 = (long)wall_clock64();
  while ((long)wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 1000;
  const int T = argc > 2 ? atoi(argv[2]) : 10;
  if (N < 1 || N > 20000 || T < 0 || T > 1000) return 2;
  const long ticks = 100L * T;
  hipStream_t s0, s1, s2;
  CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  hipEvent_t fork, j1, j2;
  CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
  CK(hipEventCreateWithFlags(&j1, hipEventDisableTiming));
  CK(hipEventCreateWithFlags(&j2, hipEventDisableTiming));
  auto chain = [&](hipStream_t s, int n) { for (int i = 0; i < n; ++i) hipLaunchKernelGGL(idle_kernel, dim3(1), dim3(64), 0, s, ticks); };
  // (0) one linear graph with 2N kernels
  hipGraph_t g; hipGraphExec_t lin, forked, ga, gb;
  CK(hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal)); chain(s0, 2 * N); CK(hipStreamEndCapture(s0, &g));
  CK(hipGraphInstantiate(&lin, g, nullptr, nullptr, 0));
  // (a) one graph, two branches of N
  CK(hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal));
  CK(hipEventRecord(fork, s0)); CK(hipStreamWaitEvent(s1, fork, 0)); CK(hipStreamWaitEvent(s2, fork, 0));
  chain(s1, N); chain(s2, N);
  CK(hipEventRecord(j1, s1)); CK(hipEventRecord(j2, s2)); CK(hipStreamWaitEvent(s0, j1, 0)); CK(hipStreamWaitEvent(s0, j2, 0));
  CK(hipStreamEndCapture(s0, &g));
  CK(hipGraphInstantiate(&forked, g, nullptr, nullptr, 0));
  // (b) two linear graphs of N
  CK(hipStreamBeginCapture(s1, hipStreamCaptureModeThreadLocal)); chain(s1, N); CK(hipStreamEndCapture(s1, &g));
  CK(hipGraphInstantiate(&ga, g, nullptr, nullptr, 0));
  CK(hipStreamBeginCapture(s2, hipStreamCaptureModeThreadLocal)); chain(s2, N); CK(hipStreamEndCapture(s2, &g));
  CK(hipGraphInstantiate(&gb, g, nullptr, nullptr, 0));
  double t[4] = {0, 0, 0, 0};
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipDeviceSynchronize()); double a = now_ms(); CK(hipGraphLaunch(lin, s0)); CK(hipStreamSynchronize(s0)); t[0] = now_ms() - a;
    a = now_ms(); CK(hipGraphLaunch(forked, s0)); CK(hipStreamSynchronize(s0)); t[1] = now_ms() - a;
    a = now_ms(); CK(hipGraphLaunch(ga, s1)); CK(hipGraphLaunch(gb, s2)); CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2)); t[2] = now_ms() - a;
    a = now_ms(); chain(s1, N); chain(s2, N); CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2)); t[3] = now_ms() - a;
  }
  printf("{\"n_per_chain\": %d, \"kernel_us\": %d, \"linear_2n_ms\": %.3f, \"forked_graph_ms\": %.3f, \"two_graphs_two_streams_ms\": %.3f, "
         "\"plain_two_streams_ms\": %.3f, \"ideal_overlap_ms\": %.3f}\n", N, T, t[0], t[1], t[2], t[3], N * T * 1e-3);
  return 0;
}
