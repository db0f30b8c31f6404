// ubench_mfma_k16.hip -- issue rate of v_mfma_f32_16x16x16_f16 (legacy K = 16 form) against v_mfma_f32_16x16x32_f16 on gfx950:
// cycles per instruction for one wave per SIMD with 4 independent accumulators.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int n) {
  f4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  h8 a8, b8; h4 a4, b4;
  for (int i = 0; i < 8; ++i) { a8[i] = (_Float16)(threadIdx.x * 0.001f + i); b8[i] = (_Float16)(1.f + i * 0.01f); }
  for (int i = 0; i < 4; ++i) { a4[i] = a8[i]; b4[i] = b8[i]; }
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < n; ++it) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (MODE == 0) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, acc[j], 0, 0, 0);
      else acc[j] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, acc[j], 0, 0, 0);
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0; for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[MODE] = t1 - t0;
}
int main() {
  float* out; unsigned long long* cyc; hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 16);
  const int n = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 2; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, out, cyc, n); else hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, out, cyc, n);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      unsigned long long c[2]; hipMemcpy(c, cyc, 16, hipMemcpyDeviceToHost);
      printf("%s: %.3f ms, %.1f ns per MFMA per wave, s_memtime ticks per MFMA %.2f\n", mode == 0 ? "16x16x32_f16" : "16x16x16_f16", ms, ms * 1e6 / (4.0 * n), (double)c[mode] / (4.0 * n));
    }
  }
  return 0;
}
