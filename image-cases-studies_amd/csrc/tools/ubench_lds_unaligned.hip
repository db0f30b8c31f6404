// ubench_lds_unaligned.hip -- does ds_read_b128 at a 2-byte-aligned LDS address work on gfx950 (unaligned access mode), and what
// does it cost?  Pattern = the B-operand window of ics_conv_mfma.hip: lane (li, lg) wants halves [8 lg - li + 15, + 8) of one row.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
typedef unsigned int u2 __attribute__((ext_vector_type(2)));

// MODE 0: one misaligned b128; 1: five aligned b64 (today's reads); 2: aligned b128 (same lanes, address rounded down to 16)
template <int MODE>
__global__ __launch_bounds__(256) void k(int iters, unsigned long long* out, unsigned* bad) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[8192];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, li = lane & 15, lg = lane >> 4;
  for (int i = tid; i < 8192; i += 256) lds[i] = (unsigned short)i;
  __syncthreads();
  const int start = 8 * lg - li + 15 + 64 * wv;   // first half
  unsigned addr = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) unsigned short*)lds + 2u * start;
  if (MODE == 1) addr &= ~7u;
  if (MODE == 2) addr &= ~15u;
  if (MODE == 3) addr &= ~3u;
  if (MODE == 4) addr = (addr & ~15u) + 8u * (lane & 1);
  u4 v = {0, 0, 0, 0};
  if (MODE != 1) {
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    if (MODE == 0) {
      bool ok = true;
      for (int d = 0; d < 4; ++d) ok = ok && v[d] == ((unsigned)(start + 2 * d) | ((unsigned)(start + 2 * d + 1) << 16));
      if (!ok) atomicAdd(bad, 1u);
      if (!ok && tid == 1) { bad[1] = v[0]; bad[2] = v[1]; bad[3] = v[2]; bad[4] = v[3]; bad[5] = start; }
    }
  }
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (MODE != 1) {
        u4 w; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(w) : "v"(addr), "n"(0)); asm volatile("s_waitcnt lgkmcnt(7)" ::: "memory");
        asm volatile("" :: "v"(w));
      } else {
        u2 w[5];
#pragma unroll
        for (int d = 0; d < 5; ++d) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(w[d]) : "v"(addr), "n"(8 * d));
        asm volatile("s_waitcnt lgkmcnt(7)" ::: "memory");
#pragma unroll
        for (int d = 0; d < 5; ++d) asm volatile("" :: "v"(w[d]));
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (lane == 0) out[blockIdx.x * 4 + wv] = t1 - t0;
}

template <int MODE>
void run(const char* name, unsigned long long* out, unsigned* bad) {
  for (int wgs = 1; wgs <= 2; ++wgs) {
    hipMemset(bad, 0, 64);
    hipLaunchKernelGGL(k<MODE>, dim3(256 * wgs), dim3(256), 0, 0, 2000, out, bad);
    hipDeviceSynchronize();
    unsigned long long h[8]; unsigned b[8];
    hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost); hipMemcpy(b, bad, sizeof b, hipMemcpyDeviceToHost);
    printf("%-44s %d WG/CU: %.1f cycles per wave-group of reads (wave view), %.1f per CU; wrong lanes %u", name, wgs, (double)h[0] / (2000 * 8.0), (double)h[0] / (2000 * 8.0 * 4 * wgs), b[0]);
    if (b[0]) printf("  (lane 1: start %u got %08x %08x %08x %08x)", b[5], b[1], b[2], b[3], b[4]);
    printf("\n");
  }
}

int main() {
  unsigned long long* out; unsigned* bad;
  hipMalloc(&out, 8 * 4 * 1024); hipMalloc(&bad, 64);
  run<0>("b128 at 2-byte alignment (B window)", out, bad);
  run<2>("b128 aligned (same lanes, rounded down)", out, bad);
  run<1>("5 x b64 aligned (today)", out, bad);
  run<3>("b128 at 4-byte alignment (parity copies)", out, bad);
  run<4>("b128 at 8-byte alignment", out, bad);
  return 0;
}
