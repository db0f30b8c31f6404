// ubench_fma.hip -- measures the fp32 VALU issue rates the convolution kernel design depends on:
// v_fmac_f32 with an SGPR weight, v_pk_fma_f32 (VGPR and SGPR-pair weights, op_sel broadcast).
// Build: hipcc --offload-arch=gfx950 -O3 ubench_fma.hip -o ubench_fma ; run on an MI355X.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));

#define REP 64
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, const float* __restrict__ w, int iters) {
  float a[16]; f2 p[8];
  const float x = out[threadIdx.x & 7];
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = x + i;
#pragma unroll
  for (int i = 0; i < 8; ++i) p[i] = (f2){x + i, x - i};
  for (int it = 0; it < iters; ++it) {
    const float w0 = w[it & 63], w1 = w[(it + 1) & 63];
    const f2 wp = (f2){w0, w1};
#pragma unroll
    for (int r = 0; r < REP; ++r) {
      if (MODE == 0) {  // 16 independent v_fmac_f32 acc, sgpr, vgpr
#pragma unroll
        for (int i = 0; i < 16; ++i) a[i] = __builtin_fmaf(w0, a[(i + 1) & 15] , a[i]);
      } else if (MODE == 1) {  // 8 independent v_pk_fma_f32, sgpr pair weight
#pragma unroll
        for (int i = 0; i < 8; ++i) p[i] = __builtin_elementwise_fma(wp, p[(i + 1) & 7], p[i]);
      } else if (MODE == 2) {  // pk with broadcast scalar input (op_sel) and sgpr pair weight
#pragma unroll
        for (int i = 0; i < 8; ++i) { const float s = p[(i + 1) & 7].x; p[i] = __builtin_elementwise_fma(wp, (f2){s, s}, p[i]); }
      } else if (MODE == 3) {  // pk, broadcast high half
#pragma unroll
        for (int i = 0; i < 8; ++i) { const float s = p[(i + 1) & 7].y; p[i] = __builtin_elementwise_fma(wp, (f2){s, s}, p[i]); }
      } else if (MODE == 4) {  // the form ics_small.hip's convolutions compile to: the WEIGHT is a per-lane register (from LDS), broadcast to both halves; two inputs in a VGPR pair
#pragma unroll
        for (int i = 0; i < 8; ++i) { const float wl = a[(i + r) & 15]; p[i] = __builtin_elementwise_fma((f2){wl, wl}, p[(i + 1) & 7], p[i]); }
      }
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += a[i];
#pragma unroll
  for (int i = 0; i < 8; ++i) s += p[i].x + p[i].y;
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, float* d, float* w, int waves_per_simd) {
  const int iters = 200;
  const int blocks = 256 * waves_per_simd;  // 256 threads = 4 waves = 1 per SIMD per block
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, w, 10);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, w, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double fma = (double)blocks * 256 * iters * REP * 16;
  printf("%-44s waves/SIMD=%d  %.3f ms  %.1f TFLOP/s\n", name, waves_per_simd, ms, 2 * fma / ms / 1e9);
}

int main() {
  float *d, *w; hipMalloc(&d, 256 * 256 * 16 * 4); hipMalloc(&w, 256);
  hipMemset(d, 0, 256 * 256 * 16 * 4); hipMemset(w, 0, 256);
  for (int wv = 1; wv <= 4; wv *= 2) {
    run<0>("v_fmac_f32 (sgpr weight)", d, w, wv);
    run<1>("v_pk_fma_f32 (sgpr pair weight)", d, w, wv);
    run<2>("v_pk_fma_f32 (sgpr pair, bcast lo input)", d, w, wv);
    run<3>("v_pk_fma_f32 (sgpr pair, bcast hi input)", d, w, wv);
    run<4>("v_pk_fma_f32 (VGPR weight bcast, VGPR pair in)", d, w, wv);
  }
  return 0;
}
