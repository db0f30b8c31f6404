// ubench_lds.hip -- LDS read cost of the A-fragment pattern of ics_conv_mfma.hip (16 lane rows at a row stride,
// four 16-byte column slots) against the contiguous pattern.  hipcc -O3 --offload-arch=gfx950 ubench_lds.hip -o ubench_lds
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef unsigned int u4 __attribute__((ext_vector_type(4)));

template <int WIDTH>
__global__ __launch_bounds__(256) void k(int rowb, int colb, int iters, unsigned long long* out, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, li = lane & 15, lg = lane >> 4;
  for (int i = tid; i < 40000 / 4; i += 256) reinterpret_cast<unsigned*>(lds)[i] = i;
  __syncthreads();
  unsigned addr = (unsigned)(li * rowb + lg * colb + wv * 32);
  unsigned acc = 0;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (WIDTH == 16) {
        u4 v; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(0)); asm volatile("s_waitcnt lgkmcnt(7)" ::: "memory");
        asm volatile("" :: "v"(v));
      } else {
        unsigned v; asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(addr)); asm volatile("s_waitcnt lgkmcnt(7)" ::: "memory");
        asm volatile("" :: "v"(v));
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (lane == 0) out[blockIdx.x * 4 + wv] = t1 - t0;
  if (acc == 12345u) sink[0] = acc;
}


typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
// MODE 0: 12 reads then 12 MFMAs; 1: read/MFMA interleaved 1:1; 2: MFMAs only; 3: reads only   (NB32 of the 12 reads are b32)
template <int MODE, int NB32>
__global__ __launch_bounds__(256) void kmix(int rowb, int colb, int iters, unsigned long long* out, float* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, li = lane & 15, lg = lane >> 4;
  for (int i = tid; i < 40000 / 4; i += 256) reinterpret_cast<unsigned*>(lds)[i] = 0x3c003c00u;
  __syncthreads();
  unsigned addr = (unsigned)(li * rowb + lg * colb + wv * 32);
  f4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  h8 a = {1, 1, 1, 1, 1, 1, 1, 1}, b = a;
  u4 v[12];
  for (int j = 0; j < 12; ++j) v[j] = (u4){0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0 || MODE == 3) {
#pragma unroll
      for (int j = 0; j < 12; ++j) {
        if (j < 12 - NB32) asm volatile("ds_read_b128 %0, %1" : "=v"(v[j]) : "v"(addr));
        else asm volatile("ds_read_b32 %0, %1" : "=v"(v[j][0]) : "v"(addr));
      }
    }
    if (MODE == 0 || MODE == 2) {
#pragma unroll
      for (int j = 0; j < 12; ++j) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[j & 3]) : "v"(a), "v"(b));
    }
    if (MODE == 1) {
#pragma unroll
      for (int j = 0; j < 12; ++j) {
        if (j < 12 - NB32) asm volatile("ds_read_b128 %0, %1" : "=v"(v[j]) : "v"(addr));
        else asm volatile("ds_read_b32 %0, %1" : "=v"(v[j][0]) : "v"(addr));
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[j & 3]) : "v"(a), "v"(b));
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int j = 0; j < 12; ++j) asm volatile("" :: "v"(v[j]));
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (lane == 0) out[blockIdx.x * 4 + wv] = t1 - t0;
  float s = 0; for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
  if (s == 12345.f) sink[0] = s;
}
template <int MODE, int NB32>
void runmix(const char* name, unsigned long long* out, float* sink) {
  for (int wgs = 1; wgs <= 2; ++wgs) {
    auto kern = kmix<MODE, NB32>;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipLaunchKernelGGL(kern, dim3(256 * wgs), dim3(256), 65536, 0, 160, 16, 2000, out, sink);
    hipDeviceSynchronize();
    unsigned long long h[8]; hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
    printf("%-56s %d WG/CU: %.0f cycles per step (12 reads + 12 MFMAs)\n", name, wgs, (double)h[0] / 2000);
  }
}

int main() {
  unsigned long long* out; unsigned* sink;
  hipMalloc(&out, 8 * 4 * 1024); hipMalloc(&sink, 64);
  const int iters = 2000;
  struct P { const char* name; int rowb, colb, width; } ps[] = {
    {"b128 contiguous (lane*16)", 16, 256, 16}, {"b128 rows 160 B, cols 16 B (k_conv_mfma A)", 160, 16, 16}, {"b128 rows 224 B, cols 16 B (two windows)", 224, 16, 16},
    {"b128 rows 144 B", 144, 16, 16}, {"b128 rows 176 B", 176, 16, 16}, {"b128 rows 272 B", 272, 16, 16}, {"b128 rows 64 B cols 16", 64, 16, 16}, {"b128 rows 1040 B", 1040, 16, 16},
    {"b32 contiguous", 4, 64, 4}, {"b32 broadcast rows", 0, 4, 4}};
  for (auto& p : ps) {
    for (int wgs = 1; wgs <= 2; ++wgs) {
      auto kern = p.width == 16 ? k<16> : k<4>;
      hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
      hipLaunchKernelGGL(kern, dim3(256 * wgs), dim3(256), 65536, 0, p.rowb, p.colb, iters, out, sink);
      hipDeviceSynchronize();
      unsigned long long h[8];
      hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
      printf("%-48s %d WG/CU: %.1f cycles per wave-instruction (wave view), %.1f per CU-instruction\n", p.name, wgs, (double)h[0] / (iters * 8.0), (double)h[0] / (iters * 8.0 * 4 * wgs));
    }
  }
  float* fs; hipMalloc(&fs, 64);
  runmix<2, 0>("12 MFMAs only", out, fs);
  runmix<3, 0>("12 b128 reads only", out, fs);
  runmix<3, 10>("2 b128 + 10 b32 reads only", out, fs);
  runmix<0, 0>("12 b128 reads, then 12 MFMAs", out, fs);
  runmix<1, 0>("12 x (b128 read, MFMA) interleaved", out, fs);
  runmix<0, 10>("2 b128 + 10 b32 reads, then 12 MFMAs", out, fs);
  runmix<1, 10>("(2 b128 + 10 b32) interleaved 1:1 with MFMAs", out, fs);
  runmix<0, 6>("6 b128 + 6 b32 reads, then 12 MFMAs", out, fs);
  return 0;
}
