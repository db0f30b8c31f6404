// ubench_fft_mix.hip -- what a transform-tile stage can expect from a CU: packed-fp32 add / mul issue rate, LDS b64 read / write rate, and
// whether the two overlap when 16 waves of one 1024-thread workgroup alternate between an LDS phase and an arithmetic phase the way the
// stages of ics_conv_fft.hip do (16 reads, ~120 packed operations, 16 writes).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_fft_mix.hip -o tools/ubench_fft_mix ; run on an MI355X.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v2f __attribute__((ext_vector_type(2)));

#define PITCH 136

// MODE 0: VALU only, 1: LDS only, 2: both (stage-like), 3: both with a workgroup barrier per stage; NV = packed operations per stage
typedef float v4f __attribute__((ext_vector_type(4)));
// VM = wave-level 16-byte global loads per thread and stage (from a 4 MB buffer that stays in L2), consumed one stage later
template <int MODE, int NV, bool PRIO, int VM = 0>
__global__ __launch_bounds__(1024) void k(float* out, int stages, const v4f* __restrict__ gsrc = nullptr) {
  extern __shared__ __attribute__((aligned(16))) v2f lds[];
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
  if (PRIO) { const int pr = (w >> 2) & 3; if (pr == 0) __builtin_amdgcn_s_setprio(0); else if (pr == 1) __builtin_amdgcn_s_setprio(1); else if (pr == 2) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(3); }
  for (int i = tid; i < 128 * PITCH; i += 1024) lds[i] = (v2f){(float)i, 1.f};
  __syncthreads();
  v2f v[16];
#pragma unroll
  for (int m = 0; m < 16; ++m) v[m] = (v2f){(float)(tid + m), 0.5f};
  typedef volatile __attribute__((address_space(3))) v2f* lp;
  const lp cp = (lp)(uint32_t)(uintptr_t)(lds + (w & 7) * PITCH + 64 * (w >> 3) + lane);      // stage A's column mapping
  v4f gl[VM > 0 ? VM : 1];
#pragma unroll
  for (int i = 0; i < (VM > 0 ? VM : 1); ++i) gl[i] = (v4f){0.f, 0.f, 0.f, 0.f};
  for (int s = 0; s < stages; ++s) {
    if (VM > 0) {   // use what the previous stage requested, request the next
#pragma unroll
      for (int i = 0; i < VM; ++i) { v[i & 15] += (v2f){gl[i].x + gl[i].z, gl[i].y + gl[i].w}; }
#pragma unroll
      for (int i = 0; i < VM; ++i) gl[i] = gsrc[(size_t)(((blockIdx.x * 7 + s * 3 + i) & 255) * 1024 + tid)];
    }
    if (MODE != 0) {
#pragma unroll
      for (int m = 0; m < 16; ++m) v[m] += cp[8 * m * PITCH];
    }
    if (MODE != 1) {
#pragma unroll
      for (int r = 0; r < NV / 16; ++r)
#pragma unroll
        for (int m = 0; m < 16; ++m) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(v[m]) : "v"(v[(m + 1) & 15]));
    }
    if (MODE != 0) {
#pragma unroll
      for (int m = 0; m < 16; ++m) cp[8 * m * PITCH] = v[m];
    }
    if (MODE == 3) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
  v2f a = v[0];
#pragma unroll
  for (int m = 1; m < 16; ++m) a += v[m];
  out[blockIdx.x * 1024 + tid] = a.x + a.y;
}

template <int MODE, int NV, bool PRIO, int VM = 0>
void run(const char* name, float* d, const v4f* g = nullptr) {
  const int stages = 400, blocks = 256;
  hipFuncSetAttribute((const void*)k<MODE, NV, PRIO, VM>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * PITCH * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE, NV, PRIO, VM>), dim3(blocks), dim3(1024), 128 * PITCH * 8, 0, d, 20, g);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE, NV, PRIO, VM>), dim3(blocks), dim3(1024), 128 * PITCH * 8, 0, d, stages, g);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  // per stage and CU: 16 waves x NV packed operations; 16 waves x 32 LDS instructions of 512 bytes
  printf("%-64s %.3f ms  = %.2f us per stage  (%.0f ns per wave-level packed op and SIMD, %.0f GB/s of LDS per CU)\n", name, ms, 1e3 * ms / stages,
         MODE == 1 ? 0.0 : 1e6 * ms / stages / (4.0 * NV), MODE == 0 ? 0.0 : 16.0 * 32 * 512 / (1e6 * ms / stages));
}

int main() {
  float* d; hipMalloc(&d, 256 * 1024 * 4);
  run<0, 128, false>("packed adds only, 128 per stage", d);
  run<0, 256, false>("packed adds only, 256 per stage", d);
  run<1, 128, false>("LDS only (16 reads + 16 writes of 8 bytes per lane)", d);
  run<2, 128, false>("both, 128 adds, no barrier", d);
  run<2, 128, true>("both, 128 adds, no barrier, four priorities per SIMD", d);
  run<3, 128, false>("both, 128 adds, barrier per stage", d);
  run<3, 128, true>("both, 128 adds, barrier per stage, four priorities", d);
  run<2, 256, false>("both, 256 adds, no barrier", d);
  run<3, 256, false>("both, 256 adds, barrier per stage", d);
  // a unit of ics_conv_fft.hip issues 32 - 40 sixteen-byte loads / stores per thread over its 8 LDS round trips: 4 per "stage" here
  v4f* g; hipMalloc(&g, (size_t)256 * 1024 * 16); hipMemset(g, 0, (size_t)256 * 1024 * 16);
  run<0, 128, false, 4>("packed adds + 4 global loads per thread and stage", d, g);
  run<1, 128, false, 4>("LDS + 4 global loads per thread and stage", d, g);
  run<2, 128, false, 4>("adds + LDS + 4 global loads, no barrier", d, g);
  run<3, 128, false, 4>("adds + LDS + 4 global loads, barrier per stage", d, g);
  run<1, 128, false, 8>("LDS + 8 global loads per thread and stage", d, g);
  run<0, 128, false, 8>("packed adds + 8 global loads per thread and stage", d, g);
  return 0;
}
