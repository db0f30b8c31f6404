// bench_gradk_mfma.hip -- stand-alone steady-state timing of the matrix-core PSF gradient (ics_gradk_mfma.hip) at
// 4096^2 x 3, 15x15 PSF, checked against a float64 sum over a few taps.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -I.. -I../../../include bench_gradk_mfma.hip -o bench_gradk_mfma
#include "../ics_gradk_mfma.hip"
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 4096, K = argc > 2 ? atoi(argv[2]) : 15;
  IcsGeom g = ics_make_geom(M, M, K);
  const size_t nf = ics_frame_floats(g), org = ics_origin_offset(g);
  std::vector<float> hu(nf, 0.f), he(nf, 0.f);
  srand(1);
  for (int y = 0; y < g.uM; ++y) for (int x = 0; x < 3 * g.uN; ++x) hu[org + (size_t)y * g.pitch + x] = (float)rand() / RAND_MAX;
  for (int y = 0; y < g.M; ++y) for (int x = 0; x < 3 * g.N; ++x) he[org + (size_t)(y + g.pad) * g.pitch + 3 * g.pad + x] = ((float)rand() / RAND_MAX - 0.5f) * 1e-2f;
  float *u, *e, *partial;
  const int nblocks = 512, NT = 16 * ((K + 15) / 16);
  hipMalloc(&u, nf * 4); hipMalloc(&e, nf * 4); hipMalloc(&partial, (size_t)nblocks * 3 * NT * NT * 4);
  hipMemcpy(u, hu.data(), nf * 4, hipMemcpyHostToDevice); hipMemcpy(e, he.data(), nf * 4, hipMemcpyHostToDevice);
  IcsGradkArgs a; a.e = e + org; a.u = u + org; a.partial = partial; a.geo = g;
  const int reps = getenv("ICS_BENCH_REPS") ? atoi(getenv("ICS_BENCH_REPS")) : 2000;
  for (int i = 0; i < 200; ++i) if (ics_launch_gradk_mfma(a, nblocks, 0) != hipSuccess) { printf("launch failed\n"); return 1; }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) ics_launch_gradk_mfma(a, nblocks, 0);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("gradk_mfma K=%d: %.4f ms\n", K, ms / reps);
  std::vector<float> hp((size_t)nblocks * 3 * NT * NT);
  hipMemcpy(hp.data(), partial, hp.size() * 4, hipMemcpyDeviceToHost);
  double worst = 0, scale = 0;
  for (int t = 0; t < 6; ++t) {
    const int ta = (t * 5) % K, tb = (t * 7 + 3) % K, c = t % 3;
    double got = 0; for (int b = 0; b < nblocks; ++b) got += hp[((size_t)b * 3 + c) * NT * NT + ta * NT + tb];
    double ref = 0;
    for (int y = 0; y < g.M; ++y) for (int x = 0; x < g.N; ++x)
      ref += (double)he[org + (size_t)(y + g.pad) * g.pitch + 3 * (x + g.pad) + c] * hu[org + (size_t)(y + g.pad + g.pad - ta) * g.pitch + 3 * (x + g.pad + g.pad - tb) + c];
    worst = fmax(worst, fabs(got - ref)); scale = fmax(scale, fabs(ref));
  }
  printf("max |gradk - float64| over 6 taps: %.3e (largest |ref| %.3e)\n", worst, scale);
  return 0;
}
