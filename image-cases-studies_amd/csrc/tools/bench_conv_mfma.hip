// bench_conv_mfma.hip -- stand-alone timing harness for the matrix-core convolution (ics_conv_mfma.hip) at
// 4096^2 x 3, 15x15 PSF; built in variants (-DICS_MFMA_ABLATE=mask) to see which phase bounds the kernel.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -I.. [-DICS_MFMA_ABLATE=m] bench_conv_mfma.hip -o bench_conv_mfma
#ifdef ICS_BENCH_SMALL_K   /* K <= 17 only (seconds to build): the other parts' entry points are stubs */
#define ICS_MFMA_PART 0
#endif
#include "../ics_conv_mfma.hip"
#ifdef ICS_BENCH_SMALL_K
hipError_t ics_launch_conv_mfma_part1(int, const IcsConvArgs&, hipStream_t) { return hipErrorInvalidValue; }
hipError_t ics_launch_conv_mfma_part2(int, const IcsConvArgs&, hipStream_t) { return hipErrorInvalidValue; }
hipError_t ics_launch_conv_mfma_part3(int, const IcsConvArgs&, hipStream_t) { return hipErrorInvalidValue; }
#endif
#include <stdio.h>
#include <stdlib.h>
#include <vector>

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 4096, K = argc > 2 ? atoi(argv[2]) : 15, N = argc > 3 ? atoi(argv[3]) : M;
  IcsGeom g = ics_make_geom(M, N, K);
  const size_t nf = ics_frame_floats(g), org = ics_origin_offset(g);
  std::vector<float> h(nf);
  srand(1);
  for (size_t i = 0; i < nf; ++i) h[i] = (float)rand() / RAND_MAX;
  float *in, *out, *f, *u, *ut; uint32_t* red; void* bt;
  hipMalloc(&in, nf * 4); hipMalloc(&out, nf * 4); hipMalloc(&f, nf * 4); hipMalloc(&u, nf * 4); hipMalloc(&ut, nf * 4);
  hipMalloc(&red, 1024); hipMemset(red, 0, 1024);
  for (float* p : {in, f, u, ut}) hipMemcpy(p, h.data(), nf * 4, hipMemcpyHostToDevice);
  hipMemset(out, 0, nf * 4);
  const size_t tf = ics_conv_mfma_table_floats(K);
  std::vector<_Float16> tab(tf * 2, (_Float16)0.f);
  const int rh = ((2 * (K + 17) + 3) & ~3) / 2;   // halves per weight row (MCfg::WROWB / 2)
  for (int c = 0; c < 3; ++c) for (int a = 0; a < K; ++a) for (int s = 0; s < 2; ++s) for (int hh = 0; hh < rh; ++hh) {
    const int b = hh - 7;
    const float w = (b >= 0 && b < K) ? 16384.f / (K * K) * (1.f + 0.01f * a + 0.02f * b) : 0.f;
    const _Float16 hi = (_Float16)w;
    tab[((size_t)c * K + a) * 2 * rh + 4 * (hh >> 1) + 2 * s + (hh & 1)] = s ? (_Float16)(w - (float)hi) : hi;
  }
  reinterpret_cast<float*>(tab.data())[tf - 4] = 1.f / 16384.f;
  hipMalloc(&bt, tf * 4); hipMemcpy(bt, tab.data(), tf * 4, hipMemcpyHostToDevice);
  IcsConvArgs a = {};
  a.in = in + org; a.out = out + org; a.f = f + org; a.u = u + org; a.ut = ut + org; a.red = red; a.lambd = 1.f; a.bt = bt; a.g = g;
  if (!getenv("ICS_BENCH_NO_ACC")) {   // the image in accumulator order for both tile heights (ics_image_acc.h)
    for (int k = 0; k < 2; ++k) {
      float* p = nullptr;
      hipMalloc(&p, ics_image_acc_floats(g, k ? 4 : 2) * 4);
      ics_launch_image_acc(a.f, g, k ? 4 : 2, p, 0);
      a.facc[k] = p;
    }
  }
  uint32_t* sched = nullptr; hipMalloc(&sched, 64); hipMemset(sched, 0, 64);
  if (!getenv("ICS_BENCH_NO_SCHED")) a.sched = sched;   // dynamic tile claiming as in the library (the launcher decides per launch)
  {
    int nb0 = -1, nb1 = -1;
    if (K == 15) {
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb0, k_conv_mfma<15, 0, 2, 1>, 256, MCfg<15, 2>::LDS_BYTES);
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb1, k_conv_mfma<15, 1, 2, 1>, 256, MCfg<15, 2>::LDS_BYTES);
      printf("occupancy (workgroups per CU) K=15, 32-row tiles: mode 0 %d, mode 1 %d, LDS %zu B\n", nb0, nb1, (size_t)MCfg<15, 2>::LDS_BYTES);
    }
  }
#ifdef ICS_MFMA_TRACE
  unsigned long long* trace; const size_t trace_n = (size_t)1024 * 4 * 1024;   // up to 1024 workgroups
  hipMalloc(&trace, trace_n * 8); hipMemset(trace, 0, trace_n * 8);
  hipMemcpyToSymbol(HIP_SYMBOL(ics_trace_buf), &trace, sizeof trace);
#endif
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 2; ++mode) {
    // long runs: the first milliseconds after idle are timed at ramping clocks, and a sustained loop of this kernel sits at
    // the board power cap (1.39 kW, sclk ~2.1 GHz) -- a 20-launch timing ranked the variants differently
    for (int i = 0; i < 200; ++i) if (ics_launch_conv_mfma(mode, a, 0) != hipSuccess) { printf("launch failed\n"); return 1; }
    hipEventRecord(e0);
    const int reps = getenv("ICS_BENCH_REPS") ? atoi(getenv("ICS_BENCH_REPS")) : 2000;   // long runs: sample clocks / power beside it
    for (int i = 0; i < reps; ++i) ics_launch_conv_mfma(mode, a, 0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("ablate=%d mode %d: %.4f ms\n", ICS_MFMA_ABLATE, mode, ms / reps);
#ifdef ICS_MFMA_TRACE
    {  // one more launch, traced
      hipMemset(trace, 0, trace_n * 8);
      ics_launch_conv_mfma(mode, a, 0); hipDeviceSynchronize();
      std::vector<unsigned long long> ht(trace_n);
      hipMemcpy(ht.data(), trace, trace_n * 8, hipMemcpyDeviceToHost);
      char nm[256]; snprintf(nm, sizeof nm, "%s/trace_mode%d.bin", getenv("ICS_TRACE_DIR") ? getenv("ICS_TRACE_DIR") : ".", mode);
      FILE* fp = fopen(nm, "wb");
      if (fp) {  // compact: per wave only the used entries
        for (size_t w = 0; w < trace_n / 1024; ++w) {
          const unsigned long long* t = ht.data() + w * 1024; int n = 0; while (n < 1024 && t[n]) ++n;
          if (!n) continue;
          unsigned long long hdr[2] = {w, (unsigned long long)n}; fwrite(hdr, 8, 2, fp); fwrite(t, 8, n, fp);
        }
        fclose(fp);
      }
    }
#endif
#ifdef ICS_MFMA_TIMING
    {
      unsigned long long h[11]; hipMemcpyFromSymbol(h, HIP_SYMBOL(ics_mfma_ticks), sizeof h);
      const double tiles = (double)g.tiles_x * g.tiles_y * 4 * 23;   // wave-tiles over the 23 launches of this mode
      const char* nm[10] = {"convert+sync", "mfma loop", "wait sync C", "epi operand issue", "wait sync D", "epilogue", "wait sync E", "prefetch issue", "transposes (LDS)", "-"};
      double tot = 0; for (int i = 0; i < 10; ++i) tot += (double)h[i];
      for (int i = 0; i < 9; ++i) printf("   %-18s %8.0f cycles / wave-tile  (%4.1f %%)\n", nm[i], h[i] / tiles, 100.0 * h[i] / tot);
      printf("   total %.0f cycles / wave-tile, %llu waves\n", tot / tiles, h[10]);
      unsigned long long z[11] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(ics_mfma_ticks), z, sizeof z);
    }
#endif
  }
  return 0;
}
