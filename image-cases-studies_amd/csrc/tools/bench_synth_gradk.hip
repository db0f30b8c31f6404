// bench_synth_gradk.hip -- stand-alone timing harness for the fused A11 + A13 kernel (ics_synth_gradk_mfma.hip) at
// 4096^2 x 3, 15x15 PSF; -DICS_FUSED_TIMING adds the per-phase cycle totals.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -I.. -I../../../include bench_synth_gradk.hip -o bench_synth_gradk
#include "../ics_synth_gradk_mfma.hip"
#include <stdio.h>
#include <stdlib.h>
#include <vector>

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 4096, K = argc > 2 ? atoi(argv[2]) : 15, N = argc > 3 ? atoi(argv[3]) : M;
  IcsGeom g = ics_make_geom(M, N, K);
  const size_t nf = ics_frame_floats(g), org = ics_origin_offset(g);
  std::vector<float> h(nf), hf(nf);
  srand(1);
  for (size_t i = 0; i < nf; ++i) { h[i] = 0.1f + 0.8f * (float)rand() / RAND_MAX; hf[i] = h[i] * (1.f + 0.002f * ((float)rand() / RAND_MAX - 0.5f)); }
  float *u, *f, *e, *partial; void* bt;
  hipMalloc(&u, nf * 4); hipMalloc(&f, nf * 4); hipMalloc(&e, nf * 4);
  hipMemcpy(u, h.data(), nf * 4, hipMemcpyHostToDevice); hipMemcpy(f, hf.data(), nf * 4, hipMemcpyHostToDevice); hipMemset(e, 0, nf * 4);
  const int rh = ((2 * (K + 17) + 3) & ~3) / 2;   // halves per weight row
  const size_t tf = (size_t)3 * K * 2 * (rh / 2) + 4;
  std::vector<_Float16> tab(tf * 2, (_Float16)0.f);
  for (int c = 0; c < 3; ++c) for (int a = 0; a < K; ++a) for (int s = 0; s < 2; ++s) for (int hh = 0; hh < rh; ++hh) {
    const int b = hh - 7;
    const float w = (b >= 0 && b < K) ? 16384.f / (K * K) * (1.f + 0.01f * a + 0.02f * b) : 0.f;
    const _Float16 hi = (_Float16)w;
    tab[((size_t)c * K + a) * 2 * rh + 4 * (hh >> 1) + 2 * s + (hh & 1)] = s ? (_Float16)(w - (float)hi) : hi;
  }
  reinterpret_cast<float*>(tab.data())[tf - 4] = 1.f / 16384.f;
  hipMalloc(&bt, tf * 4); hipMemcpy(bt, tab.data(), tf * 4, hipMemcpyHostToDevice);
  int cus = 256; hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  const int rs = getenv("ICS_BENCH_RS") ? atoi(getenv("ICS_BENCH_RS")) : 4;   // 4: 64-row tiles, two workgroups per CU; 2: 32-row tiles, three
  const int nblocks = getenv("ICS_BENCH_WGS") ? atoi(getenv("ICS_BENCH_WGS")) : (rs == 2 ? 3 : 2) * cus;
  hipMalloc(&partial, (size_t)nblocks * 768 * 4);
  IcsFusedArgs a = {};
  a.u = u + org; a.f = f + org; a.e_out = e + org; a.bt = bt; a.partial = partial; a.g = g;
  float* facc = nullptr;
  if (!getenv("ICS_BENCH_NO_ACC")) { hipMalloc(&facc, ics_image_acc_floats(g, rs) * 4); ics_launch_image_acc(a.f, g, rs, facc, 0); }
  a.facc = facc; a.rs = rs;
  a.wy0 = K / 2 + 8; a.wy1 = a.wy0 + 255; a.wx0 = a.wy0; a.wx1 = a.wy1; a.store_all = 0;
  if (K == 15) {
    int nb0 = -1;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb0, k_synth_gradk<15, true>, 256, FCfg<15>::LDS_BYTES);
    int nb2 = -1;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb2, k_synth_gradk2<15, true>, 256, FCfg2<15>::LDS_BYTES);
    printf("occupancy (workgroups per CU) K=15: 64-row %d (LDS %zu B), 32-row %d (LDS %zu B)\n", nb0, (size_t)FCfg<15>::LDS_BYTES, nb2, (size_t)FCfg2<15>::LDS_BYTES);
  }
#ifdef ICS_FUSED_TRACE
  unsigned long long* trace; const size_t trace_n = (size_t)1024 * 4 * 1024;   // up to 1024 workgroups
  hipMalloc(&trace, trace_n * 8); hipMemset(trace, 0, trace_n * 8);
  hipMemcpyToSymbol(HIP_SYMBOL(ics_fused_trace_buf), &trace, sizeof trace);
#endif
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 200; ++i) if (ics_launch_synth_gradk(a, nblocks, 0) != hipSuccess) { printf("launch failed\n"); return 1; }
  hipDeviceSynchronize();
#ifdef ICS_FUSED_TIMING
  { unsigned long long z[17] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(ics_fused_ticks), z, sizeof z); }
#endif
  hipEventRecord(e0);
  const int reps = getenv("ICS_BENCH_REPS") ? atoi(getenv("ICS_BENCH_REPS")) : 1000;
  for (int i = 0; i < reps; ++i) ics_launch_synth_gradk(a, nblocks, 0);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("synth_gradk %dx%d K=%d, %d workgroups: %.4f ms\n", M, N, K, nblocks, ms / reps);
#ifdef ICS_FUSED_TRACE
  {  // one more launch, traced (scripts/trace_conv_mfma.py, scripts/dbg/trace_tail.py)
    hipMemset(trace, 0, trace_n * 8);
    ics_launch_synth_gradk(a, nblocks, 0); hipDeviceSynchronize();
    std::vector<unsigned long long> ht(trace_n);
    hipMemcpy(ht.data(), trace, trace_n * 8, hipMemcpyDeviceToHost);
    char nm[256]; snprintf(nm, sizeof nm, "%s/trace_fused.bin", getenv("ICS_TRACE_DIR") ? getenv("ICS_TRACE_DIR") : ".");
    FILE* fp = fopen(nm, "wb");
    if (fp) {
      for (size_t w = 0; w < trace_n / 1024; ++w) {
        const unsigned long long* t = ht.data() + w * 1024; int n = 0; while (n < 1024 && t[n]) ++n;
        if (!n) continue;
        unsigned long long hdr[2] = {w, (unsigned long long)n}; fwrite(hdr, 8, 2, fp); fwrite(t, 8, n, fp);
      }
      fclose(fp);
    }
  }
#endif
#ifdef ICS_FUSED_TIMING
  {
    unsigned long long t[17]; hipMemcpyFromSymbol(t, HIP_SYMBOL(ics_fused_ticks), sizeof t);
    const double tiles = (double)((N + 63) / 64) * ((M + 63) / 64) * 4 * reps;   // wave-tiles
    const char* nm[16] = {"S0 max + barrier", "convert(0) + barrier", "conv x3", "residual x3", "B2 barrier x3", "write_e + convert x3", "B3 barrier x3", "gradk x3",
                          "f(2) + prefetch issue", "", "", "", "", "", "", ""};
    double tot = 0; for (int i = 0; i < 16; ++i) tot += (double)t[i];
    for (int i = 0; i < 9; ++i) printf("   %-24s %8.0f cycles / wave-tile  (%4.1f %%)\n", nm[i], t[i] / tiles, 100.0 * t[i] / tot);
    printf("   total %.0f cycles / wave-tile, %llu waves\n", tot / tiles, t[16]);
  }
#endif
  return 0;
}
