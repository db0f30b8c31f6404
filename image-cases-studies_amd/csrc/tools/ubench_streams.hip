// ubench_streams.hip -- does the update pass (four frames read, one written, lib/deconvolution.pyx:499-552) lose bandwidth because its five
// frames start at the same offset modulo the memory channels' interleave?  Five arrays of a frame's size from separate hipMalloc calls
// (2 MiB aligned), streamed the way k_update_rows does (16-byte accesses, streaming loads, persistent workgroups), with the arrays' starts
// skewed by k x `skew` bytes.    ./ubench_streams [frame MB] ...
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k(const f4* __restrict__ a, const f4* __restrict__ b, const f4* __restrict__ c, const f4* __restrict__ d, f4* __restrict__ o, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const f4 x = __builtin_nontemporal_load(a + i), y = __builtin_nontemporal_load(b + i), z = __builtin_nontemporal_load(c + i), w = __builtin_nontemporal_load(d + i);
    o[i] = x + y * z - w;
  }
}

int main(int argc, char** argv) {
  const int skews[] = {0, 4096, 65536 + 4096, 256 * 1024 + 8192, 1 << 20};
  for (int ai = 1; ai < (argc > 1 ? argc : 2); ++ai) {
    const size_t mb = argc > 1 ? (size_t)atol(argv[ai]) : 453, bytes = mb << 20, n = bytes / 16;
    char* p[5];
    for (int i = 0; i < 5; ++i) { if (hipMalloc((void**)&p[i], bytes + (8 << 20)) != hipSuccess) { printf("alloc failed\n"); return 1; } (void)hipMemset(p[i], 0, bytes + (8 << 20)); }
    for (int s : skews)
      for (int wg = 2; wg <= 8; wg *= 2) {
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        const f4 *a = (const f4*)(p[0]), *b = (const f4*)(p[1] + s), *c = (const f4*)(p[2] + 2 * s), *d = (const f4*)(p[3] + 3 * s);
        f4* o = (f4*)(p[4] + 4 * s);
        for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k, dim3(256 * wg), dim3(256), 0, 0, a, b, c, d, o, n);
        (void)hipEventRecord(e0);
        for (int r = 0; r < 20; ++r) hipLaunchKernelGGL(k, dim3(256 * wg), dim3(256), 0, 0, a, b, c, d, o, n);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%4zu MB per frame, skew %7d B, %d workgroups per CU: %.3f ms  %.0f GB/s\n", mb, s, wg, ms / 20, 5.0 * bytes / (ms / 20) / 1e6);
      }
    for (int i = 0; i < 5; ++i) (void)hipFree(p[i]);
  }
  return 0;
}
