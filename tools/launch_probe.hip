// Host-side cost of getting one evaluation (two dependent kernels with ~2 KB of kernel arguments each) onto the
// GPU: plain launches against a pre-instantiated two-node graph.  The queue is held by a spinning kernel while the
// host enqueues, so the numbers are host time only.   hipcc --offload-arch=gfx950 -O2 -o launch_probe launch_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define OK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("%s -> %s\n", #e, hipGetErrorString(r_)); return 1; } } while (0)
struct Big { double a[280]; };   // 2240 B, the size of PcPhaseArgs
__global__ void k_big(Big b, double* out) { if (b.a[0] == 1.25e300) out[0] = b.a[1]; }
__global__ void k_spin(long long cycles, double* out) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < cycles) {}
  if (cycles < 0) out[0] = 1.0;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  hipStream_t st; OK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  double* d; OK(hipMalloc(&d, 64));
  Big b{}; const int N = 1000;
  const long long spin = 100000000LL / 1000 * 30;   // ~30 ms at the 100 MHz wall clock
  for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(k_big, 1, 64, 0, st, b, d);
  OK(hipStreamSynchronize(st));
  // (a) 2 plain launches per evaluation
  hipLaunchKernelGGL(k_spin, 1, 1, 0, st, spin, d);
  double t0 = now();
  const int NA = 150;   // stays below the queue depth: the enqueue never blocks on the spinning kernel
  for (int i = 0; i < NA; ++i) { hipLaunchKernelGGL(k_big, 160, 64, 0, st, b, d); hipLaunchKernelGGL(k_big, 1, 256, 0, st, b, d); }
  double ta = (now() - t0) / NA;
  OK(hipStreamSynchronize(st));
  // (b) two-node graph, captured once
  hipGraph_t g; hipGraphExec_t ge;
  OK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  hipLaunchKernelGGL(k_big, 160, 64, 0, st, b, d); hipLaunchKernelGGL(k_big, 1, 256, 0, st, b, d);
  OK(hipStreamEndCapture(st, &g));
  OK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int i = 0; i < 20; ++i) OK(hipGraphLaunch(ge, st));
  OK(hipStreamSynchronize(st));
  hipLaunchKernelGGL(k_spin, 1, 1, 0, st, spin, d);
  t0 = now();
  for (int i = 0; i < N; ++i) hipGraphLaunch(ge, st);
  double tb = (now() - t0) / N;
  OK(hipStreamSynchronize(st));
  // (c) the same graph with the kernel arguments of one node replaced before every launch
  hipGraphNode_t nodes[4]; size_t nn = 4; OK(hipGraphGetNodes(g, nodes, &nn));
  hipKernelNodeParams kp; OK(hipGraphKernelNodeGetParams(nodes[nn - 1], &kp));
  hipLaunchKernelGGL(k_spin, 1, 1, 0, st, spin, d);
  t0 = now();
  for (int i = 0; i < N; ++i) { hipGraphExecKernelNodeSetParams(ge, nodes[nn - 1], &kp); hipGraphLaunch(ge, st); }
  double tc = (now() - t0) / N;
  OK(hipStreamSynchronize(st));
  // GPU side: back-to-back evaluations behind the blocker, events on the stream
  hipEvent_t e0, e1; OK(hipEventCreate(&e0)); OK(hipEventCreate(&e1));
  float ms_a, ms_b;
  hipLaunchKernelGGL(k_spin, 1, 1, 0, st, spin, d);
  OK(hipEventRecord(e0, st));
  for (int i = 0; i < N; ++i) { hipLaunchKernelGGL(k_big, 160, 64, 0, st, b, d); hipLaunchKernelGGL(k_big, 1, 256, 0, st, b, d); }
  OK(hipEventRecord(e1, st)); OK(hipStreamSynchronize(st)); OK(hipEventElapsedTime(&ms_a, e0, e1));
  hipLaunchKernelGGL(k_spin, 1, 1, 0, st, spin, d);
  OK(hipEventRecord(e0, st));
  for (int i = 0; i < N; ++i) hipGraphLaunch(ge, st);
  OK(hipEventRecord(e1, st)); OK(hipStreamSynchronize(st)); OK(hipEventElapsedTime(&ms_b, e0, e1));
  printf("host us per evaluation: 2 launches %.2f | graph launch %.2f | graph + 1 SetParams %.2f\n", ta * 1e6, tb * 1e6, tc * 1e6);
  printf("GPU us per evaluation (2 empty kernels, queued): launches %.2f | graph %.2f\n", ms_a * 1e3 / N, ms_b * 1e3 / N);
  return 0;
}
