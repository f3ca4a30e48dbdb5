// Does a raw buffer store range-check per dword on gfx950?  flush_run's candidate replacement: 16-byte stores through a
// buffer descriptor of exactly `len` doubles, no lane predicates, the odd last element and the partial last batch left to
// the hardware.  Copies src[0 .. len) to dst + shift for every len in [0, 1100) and shift in {0, 1} (8- and 16-byte aligned
// starts) and checks that exactly those doubles changed.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/bufstore_probe tools/bufstore_probe.hip && /tmp/bufstore_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef double d2 __attribute__((ext_vector_type(2), aligned(8)));
typedef int i4 __attribute__((ext_vector_type(4)));

__global__ void copy_run(double* dst, const double* src, int len) {
  extern __shared__ double lds[];
  const int tid = threadIdx.x;
  for (int i = tid; i < 2304; i += 64) lds[i] = i < len ? src[i] : -777.0;
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(dst, 0, len * 8, 0x00020000);
  const int pairs = (len + 1) >> 1;
  int voff = 16 * tid;
  for (int b0 = 0; b0 < pairs; b0 += 4 * 64, voff += 4 * 64 * 16) {
    d2 a[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) a[q] = *reinterpret_cast<const d2*>(lds + 2 * (b0 + q * 64 + tid));
#pragma unroll
    for (int q = 0; q < 4; ++q) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i4, a[q]), r, voff + 1024 * q, 0, 0);
  }
}

int main() {
  const int CAP = 1400;
  double *d_src, *d_dst;
  hipMalloc(&d_src, CAP * 8);
  hipMalloc(&d_dst, CAP * 8);
  std::vector<double> src(CAP), out(CAP), canary(CAP, -1.0);
  for (int i = 0; i < CAP; ++i) src[i] = 1000.0 + i;
  hipMemcpy(d_src, src.data(), CAP * 8, hipMemcpyHostToDevice);
  int bad = 0;
  for (int shift = 0; shift < 2; ++shift)
    for (int len = 0; len < 1100; ++len) {
      hipMemcpy(d_dst, canary.data(), CAP * 8, hipMemcpyHostToDevice);
      hipLaunchKernelGGL(copy_run, dim3(1), dim3(64), 2304 * 8, 0, d_dst + shift, d_src, len);
      hipMemcpy(out.data(), d_dst, CAP * 8, hipMemcpyDeviceToHost);
      for (int i = 0; i < CAP; ++i) {
        const double want = (i >= shift && i < shift + len) ? src[i - shift] : -1.0;
        if (out[i] != want) {
          if (bad < 10) std::printf("shift %d len %d: [%d] = %g, want %g\n", shift, len, i, out[i], want);
          ++bad;
        }
      }
    }
  std::printf(bad ? "FAILED: %d wrong doubles\n" : "ok: every run copied exactly (per-dword range check holds)\n", bad);
  return bad != 0;
}
