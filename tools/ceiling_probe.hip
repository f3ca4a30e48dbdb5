// Hand-written ceilings for a kernel that writes 90 % of its bytes (Weak 11): what the device sustains
// for the evaluation's OWN launch shape -- as many workgroups of as many threads, every workgroup streaming its
// contiguous share with 16-byte stores -- instead of torch.fill_'s shape.  Two kernels:
//   store_only   every lane stores 16 B per instruction, workgroup b writes [b * share, (b + 1) * share)
//   read10_write90   the same stores, fed by loads of a tenth as many bytes (the evaluation reads x~ and lambda: about
//                    a tenth of what it writes) -- every store's value depends on a loaded one, as in the evaluation
// Build and run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o tools/ceiling_probe tools/ceiling_probe.hip
//                                tools/ceiling_probe <bytes> <workgroups> <threads> [launches]
// Prints the mean launch time between two HIP events around `launches` back-to-back launches (default 400) and the
// bandwidth of the written (+ read) bytes.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define OK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                 \
      std::exit(1);                                                                \
    }                                                                              \
  } while (0)

typedef double d2 __attribute__((ext_vector_type(2)));

__global__ void store_only(d2* __restrict__ out, long long pairs_per_block, double v) {
  d2* p = out + (long long)blockIdx.x * pairs_per_block;
  const d2 val = {v, v + 1.0};
  for (long long i = threadIdx.x; i < pairs_per_block; i += blockDim.x) p[i] = val;
}

__global__ void read10_write90(const d2* __restrict__ in, d2* __restrict__ out, long long pairs_per_block) {
  d2* p = out + (long long)blockIdx.x * pairs_per_block;
  const d2* q = in + (long long)blockIdx.x * (pairs_per_block / 9);
  const long long nin = pairs_per_block / 9;
  for (long long i0 = 0; i0 < nin; i0 += blockDim.x) {   // one load feeds nine stores
    const long long i = i0 + threadIdx.x;
    d2 a = {0.0, 0.0};
    if (i < nin) a = q[i];
#pragma unroll
    for (int r = 0; r < 9; ++r) {
      const long long o = (i0 * 9) + (long long)r * blockDim.x + threadIdx.x;
      if (o < pairs_per_block) p[o] = a + (double)r;
    }
  }
}

int main(int argc, char** argv) {
  if (argc < 4) {
    std::fprintf(stderr, "usage: %s <bytes> <workgroups> <threads> [launches]\n", argv[0]);
    return 2;
  }
  const long long bytes = std::atoll(argv[1]);
  const int wg = std::atoi(argv[2]), th = std::atoi(argv[3]);
  const int n = argc > 4 ? std::atoi(argv[4]) : 400;
  const long long ppb = bytes / 16 / wg;   // 16-byte pairs per workgroup
  d2 *out = nullptr, *in = nullptr;
  OK(hipMalloc(&out, (size_t)ppb * wg * 16));
  OK(hipMalloc(&in, (size_t)(ppb / 9 + 1) * wg * 16));
  OK(hipMemset(in, 0, (size_t)(ppb / 9 + 1) * wg * 16));
  hipEvent_t e0, e1;
  OK(hipEventCreate(&e0));
  OK(hipEventCreate(&e1));
  for (int which = 0; which < 2; ++which) {
    for (int i = 0; i < 20; ++i) {
      if (which == 0) store_only<<<wg, th>>>(out, ppb, 1.0);
      else read10_write90<<<wg, th>>>(in, out, ppb);
    }
    OK(hipDeviceSynchronize());
    OK(hipEventRecord(e0));
    for (int i = 0; i < n; ++i) {
      if (which == 0) store_only<<<wg, th>>>(out, ppb, (double)i);
      else read10_write90<<<wg, th>>>(in, out, ppb);
    }
    OK(hipEventRecord(e1));
    OK(hipEventSynchronize(e1));
    float ms = 0;
    OK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / n;
    const double moved = (double)ppb * wg * 16 * (which == 0 ? 1.0 : 1.0 + 1.0 / 9.0);
    std::printf("%-15s %12lld B written  %5d x %3d  %8.2f us  %6.2f TB/s (written%s)\n", which == 0 ? "store_only" : "read10_write90",
                ppb * wg * 16, wg, th, us, moved / us * 1e-6, which == 0 ? "" : " + read");
  }
  return 0;
}
