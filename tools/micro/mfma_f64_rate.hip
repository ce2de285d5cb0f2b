// Development microbenchmark: the issue rate of v_mfma_f64_16x16x4 on gfx950 — one or two wavefronts per SIMD, 4 independent
// accumulator chains each, optionally with LDS traffic beside it. hipcc --offload-arch=gfx950 -O3 mfma_f64_rate.hip -o mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));
template <int LDSOPS>
__global__ __launch_bounds__(256) void k_rate(double *out, int iters, double a0, double b0) {
  extern __shared__ d2 sm[];
  d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  double a = a0 + threadIdx.x * 1e-9, b = b0;
  d2 v = {a, b};
  for (int i = 0; i < iters; ++i) {
    if (LDSOPS > 0) {
#pragma unroll
      for (int q = 0; q < LDSOPS; ++q) sm[threadIdx.x + 256 * q] = v;
#pragma unroll
      for (int q = 0; q < LDSOPS; ++q) v += sm[(threadIdx.x ^ 1) + 256 * q];
      a = v.x;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, b, c3, 0, 0, 0);
    }
  }
  d4 s = c0 + c1 + c2 + c3;
  out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3] + v.y;
}
template <int LDSOPS>
static void run(int wgs, int iters, const char *what) {
  double *out;
  hipMalloc(&out, sizeof(double) * wgs * 256);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const size_t lds = 256 * 16 * (LDSOPS > 0 ? LDSOPS : 1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_rate<LDSOPS>, dim3(wgs), dim3(256), lds, 0, out, iters, 1e-3, 2e-3);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double mfma_per_simd = (double)wgs / 256.0 * iters * 16.0;   // a workgroup's 4 wavefronts sit on the 4 SIMDs of a CU
    if (rep == 2)
      printf("%-34s wgs=%4d iters=%5d  %8.1f us  %6.1f ns per instruction and SIMD (= %5.1f cycles at 2.4 GHz)  %5.1f TFLOP/s\n", what, wgs,
             iters, ms * 1e3, ms * 1e6 / mfma_per_simd, ms * 1e6 / mfma_per_simd * 2.4, (double)wgs * 4 * iters * 16 * 2048 / (ms * 1e-3) / 1e12);
  }
  hipFree(out);
}
int main() {
  run<0>(256, 256, "1 wave/SIMD");
  run<0>(512, 256, "2 waves/SIMD");
  run<0>(512, 64, "2 waves/SIMD, short");
  run<0>(1024, 128, "4 waves/SIMD");
  run<2>(512, 256, "2 waves/SIMD + 2 LDS w/r per 16");
  run<8>(512, 256, "2 waves/SIMD + 8 LDS w/r per 16");
  return 0;
}
