// mfma_f16_layout.hip -- checks the operand/result lane layout of v_mfma_f32_16x16x32_f16 that the
// fp16-split convolution kernel relies on, with asymmetric integer matrices (exact in fp16/fp32):
//   A[i][k]: lane l holds row i = l & 15, k = 8*(l >> 4) .. +7        (8 halves = one 128-bit register group)
//   B[k][j]: lane l holds col j = l & 15, k = 8*(l >> 4) .. +7
//   D[i][j]: lane l holds col j = l & 15, rows i = 4*(l >> 4) + r, r = 0..3
// Also times a dependent-free stream of these MFMAs (4 accumulators per wave) for the achievable rate.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void k_layout(const float* A, const float* B, float* D) {
  const int l = threadIdx.x, i = l & 15, g = l >> 4;
  h8 a, b;
  for (int t = 0; t < 8; ++t) { a[t] = (_Float16)A[i * 32 + 8 * g + t]; b[t] = (_Float16)B[(8 * g + t) * 16 + i]; }
  f4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[(4 * g + r) * 16 + i] = c[r];
}

__global__ __launch_bounds__(256) void k_rate(float* out, int iters) {
  h8 a, b;
  for (int t = 0; t < 8; ++t) { a[t] = (_Float16)(threadIdx.x * 0.001f + t); b[t] = (_Float16)(t * 0.5f - threadIdx.x * 0.002f); }
  f4 c0 = {0,0,0,0}, c1 = c0, c2 = c0, c3 = c0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c3, 0, 0, 0);
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}

int main() {
  float hA[16 * 32], hB[32 * 16], hD[256], ref[256];
  for (int i = 0; i < 16; ++i) for (int k = 0; k < 32; ++k) hA[i * 32 + k] = (float)((i * 7 + k * 3) % 11 - 5);
  for (int k = 0; k < 32; ++k) for (int j = 0; j < 16; ++j) hB[k * 16 + j] = (float)((k * 5 + j * 2) % 13 - 6);
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { float s = 0; for (int k = 0; k < 32; ++k) s += hA[i * 32 + k] * hB[k * 16 + j]; ref[i * 16 + j] = s; }
  float *dA, *dB, *dD;
  hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD);
  hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
  int bad = 0; for (int i = 0; i < 256; ++i) bad += (hD[i] != ref[i]);
  printf("layout check: %d mismatches of 256 (%s)\n", bad, bad ? "WRONG LAYOUT ASSUMPTION" : "layout confirmed");
  float* o; hipMalloc(&o, 256 * 4 * 1024 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wv = 1; wv <= 2; ++wv) {
    const int blocks = 256 * wv, iters = 2000;
    hipLaunchKernelGGL(k_rate, dim3(blocks), dim3(256), 0, 0, o, 10);
    hipEventRecord(e0); hipLaunchKernelGGL(k_rate, dim3(blocks), dim3(256), 0, 0, o, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)blocks * 4 * iters * 64 * 2.0 * 16 * 16 * 32;
    printf("mfma_f32_16x16x32_f16: %d wave(s)/SIMD  %.3f ms  %.0f TFLOP/s\n", wv, ms, flop / ms / 1e9);
  }
  return bad;
}
