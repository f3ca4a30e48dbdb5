// Diagnostic: clock held during back-to-back tiny launches and the latency of dependent global loads.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void probe(const int* __restrict__ next, long long* stamps, int nchain, int* sink) {
  long long t0 = __builtin_amdgcn_s_memtime();
  long long r0 = __builtin_amdgcn_s_memrealtime();
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  for (int k = 0; k < nchain; ++k) idx = next[idx];
  long long t1 = __builtin_amdgcn_s_memtime();
  long long r1 = __builtin_amdgcn_s_memrealtime();
  if (idx == -12345) sink[0] = idx;
  if (threadIdx.x == 0) {
    stamps[blockIdx.x * 4 + 0] = t0; stamps[blockIdx.x * 4 + 1] = t1;
    stamps[blockIdx.x * 4 + 2] = r0; stamps[blockIdx.x * 4 + 3] = r1;
  }
}
__global__ void empty_k(int* sink) { if (threadIdx.x == 12345) sink[0] = 1; }

int main() {
  const int blocks = 167, tb = 64, n = blocks * tb;
  std::vector<int> h(n);
  for (int i = 0; i < n; ++i) h[i] = (i * 7919 + 13) % n;   // scattered chain, 4-byte elements over 42 KB
  int *d_next, *d_sink; long long* d_st;
  CK(hipMalloc(&d_next, n * 4)); CK(hipMalloc(&d_sink, 4)); CK(hipMalloc(&d_st, blocks * 4 * 8));
  CK(hipMemcpy(d_next, h.data(), n * 4, hipMemcpyHostToDevice));
  hipStream_t s; CK(hipStreamCreate(&s));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int nchain : {0, 1, 2, 4, 8}) {
    for (int i = 0; i < 200; ++i) probe<<<blocks, tb, 0, s>>>(d_next, d_st, nchain, d_sink);
    CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s));
    const int reps = 3000;
    for (int i = 0; i < reps; ++i) probe<<<blocks, tb, 0, s>>>(d_next, d_st, nchain, d_sink);
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<long long> st(blocks * 4);
    CK(hipMemcpy(st.data(), d_st, blocks * 32, hipMemcpyDeviceToHost));
    double cyc = 0, real = 0;
    for (int b = 0; b < blocks; ++b) { cyc += st[b * 4 + 1] - st[b * 4]; real += st[b * 4 + 3] - st[b * 4 + 2]; }
    cyc /= blocks; real /= blocks;
    printf("nchain %d: %.2f us/launch (host-paired), in-kernel %.0f cycles, %.0f ns, clock %.2f GHz\n", nchain,
           ms * 1e3 / reps, cyc, real * 10.0, real > 0 ? cyc / (real * 10.0) : 0.0);
  }
  CK(hipEventRecord(e0, s));
  for (int i = 0; i < 3000; ++i) empty_k<<<blocks, tb, 0, s>>>(d_sink);
  CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("empty kernel: %.2f us/launch\n", ms * 1e3 / 3000);
  return 0;
}
