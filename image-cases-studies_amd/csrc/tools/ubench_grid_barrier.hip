// ubench_grid_barrier.hip -- what a grid-wide barrier costs on this part, and what a cooperative launch costs on top of a plain one: the two
// prices of an iteration kernel for small frames (ics_small.hip) that keeps its tiles in LDS and meets its neighbours through global memory.
// Every workgroup writes a 4 KB slab, all meet at the barrier, every workgroup reads its neighbour's slab and checks it.
//   variant 0  plain stores / loads, __threadfence() either side of the counter (release: L2 write-back, acquire: L2 invalidate)
//   variant 1  the slabs go through agent-scope relaxed atomics (coherent accesses), the barrier is the counter alone
//   variant 2  no data, the counter alone
//   ./ubench_grid_barrier [workgroups] [threads] [barriers per launch]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int V>
__device__ __forceinline__ void grid_barrier(unsigned long long* ctr, unsigned long long target) {
  __syncthreads();
  if (threadIdx.x == 0) {
    if (V == 0) __threadfence();
    __hip_atomic_fetch_add(ctr, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    if (V == 0) __threadfence();
  }
  __syncthreads();
}

template <int V>
__global__ __launch_bounds__(1024) void k(float* buf, unsigned long long* ctr, unsigned long long base, int nb, int* bad) {
  const int w = blockIdx.x, nw = gridDim.x, t = threadIdx.x, nt = blockDim.x;
  float* mine = buf + (size_t)w * 1024;
  const float* theirs = buf + (size_t)((w + 97) % nw) * 1024;
  int wrong = 0;
  for (int b = 0; b < nb; ++b) {
    const float tag = (float)(base % 1000ull) + (float)b;
    if (V == 0) for (int i = t; i < 1024; i += nt) mine[i] = tag + (float)i;
    if (V == 1) for (int i = t; i < 1024; i += nt) __hip_atomic_store(mine + i, tag + (float)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    grid_barrier<V>(ctr, (base + (unsigned long long)b) * nw + nw);
    if (V == 0) for (int i = t; i < 1024; i += nt) wrong += theirs[i] != tag + (float)i;
    if (V == 1) for (int i = t; i < 1024; i += nt) wrong += __hip_atomic_load(theirs + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != tag + (float)i;
    // (the next round's stores must not overtake a neighbour's reads of this round: the barrier of the next round is preceded by a second one
    //  in a real kernel; here the slabs are double-buffered by parity instead)
    mine += (b & 1) ? -(ptrdiff_t)(nw * 1024) : (ptrdiff_t)(nw * 1024);
    theirs += (b & 1) ? -(ptrdiff_t)(nw * 1024) : (ptrdiff_t)(nw * 1024);
  }
  if (wrong) atomicAdd(bad, wrong);
}
__global__ void k_empty(int* p) { if (p && threadIdx.x == 9999) *p = 1; }

template <int V>
static void run(int nw, int nt, int nb, bool coop, float* buf, unsigned long long* ctr, int* bad) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  unsigned long long base = 0; (void)hipMemset(ctr, 0, 8); (void)hipMemset(bad, 0, 4);
  const int reps = 50;
  float best = 1e30f;
  for (int pass = 0; pass < 3; ++pass) {
    (void)hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) {
      if (coop) {
        void* args[] = {&buf, &ctr, &base, &nb, &bad};
        hipError_t e = hipLaunchCooperativeKernel(reinterpret_cast<const void*>(k<V>), dim3(nw), dim3(nt), args, 0, 0);
        if (e != hipSuccess) { printf("cooperative launch failed: %s\n", hipGetErrorString(e)); return; }
      } else hipLaunchKernelGGL(k<V>, dim3(nw), dim3(nt), 0, 0, buf, ctr, base, nb, bad);
      base += nb;
    }
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  int hb = 0; (void)hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
  printf("variant %d  %s launch  %3d workgroups x %4d threads  %2d barriers per launch: %7.2f us per launch  (%d wrong values)\n", V, coop ? "cooperative" : "plain      ", nw, nt, nb, best / reps * 1e3f, hb);
}

int main(int argc, char** argv) {
  const int nw = argc > 1 ? atoi(argv[1]) : 243, nt = argc > 2 ? atoi(argv[2]) : 512;
  float* buf; unsigned long long* ctr; int* bad;
  (void)hipMalloc((void**)&buf, (size_t)2 * nw * 1024 * 4); (void)hipMalloc((void**)&ctr, 8); (void)hipMalloc((void**)&bad, 4);
  {  // dependent empty launches: the floor a multi-launch iteration pays per kernel
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(k_empty, dim3(nw), dim3(nt), 0, 0, bad);
    (void)hipEventRecord(e0);
    for (int r = 0; r < 200; ++r) hipLaunchKernelGGL(k_empty, dim3(nw), dim3(nt), 0, 0, bad);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("empty kernel, back to back on one stream: %.2f us per launch\n", ms / 200 * 1e3f);
  }
  const int nbs[] = {1, 5, 21};
  for (int coop = 0; coop < 2; ++coop)
    for (int nb : nbs) {
      if (argc > 3 && nb != atoi(argv[3])) continue;
      run<0>(nw, nt, nb, coop, buf, ctr, bad);
      run<1>(nw, nt, nb, coop, buf, ctr, bad);
      run<2>(nw, nt, nb, coop, buf, ctr, bad);
    }
  return 0;
}
