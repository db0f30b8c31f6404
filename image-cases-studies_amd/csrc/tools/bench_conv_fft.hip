// bench_conv_fft.hip -- stand-alone harness for the overlap-save FFT convolution (ics_conv_fft.hip).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -I.. -I../../../include bench_conv_fft.hip -o bench_conv_fft
//   ./bench_conv_fft emulate [M K N]     CPU emulation of the kernel's stages (host pass of the same functions, "threads" run one after
//                                        the other per stage) against float64 direct sums over the whole output: no GPU needed
//   ./bench_conv_fft M K [N] [reps]      GPU: both modes checked against float64 direct sums on sampled rows, then timed
#include "../ics_conv_fft.hip"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

namespace {

int g_planar = 1;   // every frame the kernel touches is a channel-planar mirror (ics_common.h); the HWC form of round 5's first versions is gone

struct Host {
  IcsGeom g;
  size_t nf, org;
  std::vector<float> u, e, f, ut, psf;   // frames (u-frame geometry) and the K x K x 3 PSF
  std::vector<float> pu, pe, pf, put;    // their planar mirrors
  size_t pnf, porg;
};

void to_planar(const Host& h, const std::vector<float>& src, std::vector<float>& dst) {
  const IcsGeom& g = h.g; const int pp = ics_ppitch(g); const size_t pl = ics_plane_floats(g);
  dst.assign(h.pnf, 0.f);
  for (int y = -g.ay; y < g.rows - g.ay; ++y)
    for (int x = -g.ax; x < pp - g.ax && 3 * (x + g.ax) + 2 < g.pitch; ++x)
      for (int c = 0; c < 3; ++c)
        dst[c * pl + (size_t)(y + g.ay) * pp + x + g.ax] = src[(size_t)(y + g.ay) * g.pitch + 3 * (x + g.ax) + c];
}
void from_planar(const Host& h, const std::vector<float>& src, std::vector<float>& dst) {
  const IcsGeom& g = h.g; const int pp = ics_ppitch(g); const size_t pl = ics_plane_floats(g);
  for (int y = 0; y < g.uM; ++y)
    for (int x = 0; x < g.uN; ++x)
      for (int c = 0; c < 3; ++c)
        dst[h.org + (size_t)y * g.pitch + 3 * x + c] = src[c * pl + h.porg + (size_t)y * pp + x];
}

Host make_host(int M, int N, int K) {
  Host h;
  h.g = ics_make_geom(M, N, K);
  h.nf = ics_frame_floats(h.g); h.org = ics_origin_offset(h.g);
  h.u.assign(h.nf, 0.f); h.e.assign(h.nf, 0.f); h.f.assign(h.nf, 0.f); h.ut.assign(h.nf, 0.f);
  srand(7);
  const IcsGeom& g = h.g;
  for (int y = 0; y < g.uM; ++y)
    for (int x = 0; x < g.uN; ++x)
      for (int c = 0; c < 3; ++c) {
        const size_t o = h.org + (size_t)y * g.pitch + 3 * x + c;
        h.u[o] = 0.1f + 0.8f * (float)rand() / RAND_MAX;
        h.ut[o] = h.u[o] + 1e-3f * ((float)rand() / RAND_MAX - 0.5f);
        const bool in = y >= g.pad && y < g.pad + M && x >= g.pad && x < g.pad + N;
        h.e[o] = in ? 2e-3f * ((float)rand() / RAND_MAX - 0.5f) : 0.f;
        h.f[o] = in ? 0.1f + 0.8f * (float)rand() / RAND_MAX : 0.f;
      }
  h.psf.resize((size_t)K * K * 3);
  double sum[3] = {0, 0, 0};
  for (int a = 0; a < K; ++a)
    for (int b = 0; b < K; ++b)
      for (int c = 0; c < 3; ++c) {
        const double s = K / 6.0 * (1.0 + 0.1 * c), dy = a - K / 2 + 0.3, dx = b - K / 2 - 0.2 * c;   // not symmetric: orientation errors show
        const double v = exp(-(dy * dy + dx * dx) / (2 * s * s)) * (1.0 + 0.05 * ((a * 7 + b * 3) % 5));
        h.psf[((size_t)a * K + b) * 3 + c] = (float)v; sum[c] += v;
      }
  for (size_t i = 0; i < h.psf.size(); ++i) h.psf[i] = (float)(h.psf[i] / sum[i % 3]);
  h.pnf = ics_planar_floats(h.g); h.porg = ics_planar_origin(h.g);
  to_planar(h, h.u, h.pu); to_planar(h, h.e, h.pe); to_planar(h, h.f, h.pf); to_planar(h, h.ut, h.put);
  return h;
}

// float64 direct sum of one output value in u-frame coordinates
double direct(const Host& h, int mode, int y, int x, int c) {
  const IcsGeom& g = h.g;
  const std::vector<float>& in = mode == 0 ? h.u : h.e;
  double s = 0.0;
  for (int a = 0; a < g.K; ++a)
    for (int b = 0; b < g.K; ++b) {
      const double w = mode == 0 ? h.psf[((size_t)(g.K - 1 - a) * g.K + (g.K - 1 - b)) * 3 + c] : h.psf[((size_t)a * g.K + b) * 3 + c];
      const ptrdiff_t o = (ptrdiff_t)h.org + (ptrdiff_t)(y + a - g.pad) * g.pitch + 3 * (x + b - g.pad) + c;
      s += w * (double)in[o];
    }
  return s;
}

void host_spectrum(const Host& h, int o, std::vector<v2f>& spec) {
  const int K = h.g.K;
  spec.assign((size_t)3 * 128 * 128, (v2f){0.f, 0.f});
  std::vector<double> G((size_t)K * 128 * 2);
  for (int c = 0; c < 3; ++c) {
    for (int a = 0; a < K; ++a)
      for (int kx = 0; kx < 128; ++kx) {
        double re = 0, im = 0;
        for (int b = 0; b < K; ++b) {
          const double w = o == 0 ? h.psf[((size_t)(K - 1 - a) * K + (K - 1 - b)) * 3 + c] : h.psf[((size_t)a * K + b) * 3 + c];
          const double ph = -2.0 * M_PI * ((b * kx) & 127) / 128.0;
          re += w * cos(ph); im += w * sin(ph);
        }
        G[((size_t)a * 128 + kx) * 2] = re; G[((size_t)a * 128 + kx) * 2 + 1] = im;
      }
    for (int ky = 0; ky < 128; ++ky)
      for (int kx = 0; kx < 128; ++kx) {
        double re = 0, im = 0;
        for (int a = 0; a < K; ++a) {
          const double ph = -2.0 * M_PI * ((a * ky) & 127) / 128.0, wr = cos(ph), wi = sin(ph);
          const double gr = G[((size_t)a * 128 + kx) * 2], gi = G[((size_t)a * 128 + kx) * 2 + 1];
          re += gr * wr - gi * wi; im += gr * wi + gi * wr;
        }
        spec[icsfft::spec_index(c, ky, kx)] = (v2f){(float)(re / 16384.0), (float)(-im / 16384.0)};
      }
  }
}

IcsConvArgs conv_args(const Host& h, int mode, const float* in, float* out, const float* f, const float* u, const float* ut, uint32_t* red) {
  IcsConvArgs a = {};
  const size_t org = g_planar ? h.porg : h.org;
  a.in = in + org; a.out = out + org; a.f = f + org; a.u = u + org; a.ut = ut + org; a.red = red; a.lambd = 10000.f; a.g = h.g;
  a.tv = nullptr; a.tv_kind = 0;
  return a;
}

double check(const Host& h, int mode, const std::vector<float>& out, int row_step, double* worst_abs) {
  const IcsGeom& g = h.g;
  double worst = 0, ref_max = 0;
  int nbad = 0;
  const int y0 = mode == 0 ? g.pad : 0, y1 = mode == 0 ? g.pad + g.M : g.uM, x0 = mode == 0 ? g.pad : 0, x1 = mode == 0 ? g.pad + g.N : g.uN;
  for (int y = y0; y < y1; y += (y < y0 + 3 || y >= y1 - 3) ? 1 : row_step)
    for (int x = x0; x < x1; ++x)
      for (int c = 0; c < 3; ++c) {
        const size_t o = h.org + (size_t)y * g.pitch + 3 * x + c;
        double r = direct(h, mode, y, x, c);
        ref_max = fmax(ref_max, fabs(r));
        const double conv = r;
        if (mode == 0) r -= (double)h.f[o];
        if (getenv("ICS_FFT_DEBUG") && fabs(r - (double)out[o]) > 1e-3 && nbad++ < 200000)
          fprintf(stderr, "%d %d %d %d %.6f %.6f %.6f %.6f\n", mode, y, x, c, out[o], r, conv, h.f[o]);
        worst = fmax(worst, fabs(r - (double)out[o]));
      }
  *worst_abs = worst;
  return worst / ref_max;   // relative to the largest convolution value (the stage gate of tests/test_gpu_stages.py)
}

int emulate(int M, int K, int N) {
  Host h = make_host(M, N, K);
  std::vector<v2f> lds((size_t)ICS_FFT_P * ICS_FFT_PITCH + 128);
  int rc = 0;
  for (int mode = 0; mode < 2; ++mode) {
    std::vector<v2f> spec;
    host_spectrum(h, mode, spec);
    std::vector<float> out(h.nf, 0.f), pout(g_planar ? h.pnf : 0, 0.f);
    uint32_t red[16] = {0};
    IcsFftArgs a;
    if (g_planar) ics_conv_fft_fill_args(mode, conv_args(h, mode, mode == 0 ? h.pu.data() : h.pe.data(), pout.data(), h.pf.data(), h.pu.data(), h.put.data(), red), (const float*)spec.data(), &a);
    else ics_conv_fft_fill_args(mode, conv_args(h, mode, mode == 0 ? h.u.data() : h.e.data(), out.data(), h.f.data(), h.u.data(), h.ut.data(), red), (const float*)spec.data(), &a);
    a.planar = 63;
    printf("mode %d: V %d x %d, tiles %d (x %d), units %d\n", mode, a.Vy, a.V, a.ntiles, a.tiles_x, a.nunits);
    const icsfft::Mem mem = icsfft::make_mem(a);
    std::vector<v2f> twl(ICS_FFT_TW_ENTRIES);
    for (int t = 0; t < ICS_FFT_TW_ENTRIES; ++t) twl[t] = icsfft::tw128((t / ICS_FFT_TWS) * (t % ICS_FFT_TWS));
    for (int n = 0; n < a.nunits; ++n) {
      const icsfft::Unit u = icsfft::decode_unit(a, n);
      for (int t = 0; t < 1024; ++t) { v4f pw[2][4]; icsfft::load_window(a, mem, u, t, pw); icsfft::store_window(pw, lds.data(), t); }
      for (int t = 0; t < 1024; ++t) icsfft::stage_a(lds.data(), t);
      for (int t = 0; t < 1024; ++t) icsfft::stage_b<1>(lds.data(), t);
      // C, D, E exchange inside a wave that runs in lock step (reads of all lanes before the writes): C and E read a snapshot here
      { const std::vector<v2f> snap = lds; for (int t = 0; t < 1024; ++t) icsfft::stage_c(snap.data(), lds.data(), twl.data(), t); }
      for (int t = 0; t < 1024; ++t) { v2f sp[2][8]; icsfft::load_spectrum(mem, u.c, t, sp); icsfft::stage_d(sp, lds.data(), t); }
      { const std::vector<v2f> snap = lds; for (int t = 0; t < 1024; ++t) icsfft::stage_e(snap.data(), lds.data(), twl.data(), t); }
      for (int t = 0; t < 1024; ++t) icsfft::stage_b<-1>(lds.data(), t);
      for (int t = 0; t < 1024; ++t) icsfft::stage_g(lds.data(), t);
      for (int t = 0; t < 1024; ++t) {
        icsfft::Maxima mx; icsfft::maxima_init(mx);
        v4f fimg[2][4];
        icsfft::Ops o;
        if (mode == 0) icsfft::load_image(a, mem, u, t, fimg);
        else { icsfft::load_ops<true>(a, mem, u, t, 0, o); icsfft::load_ops<true>(a, mem, u, t, 1, o); }
        icsfft::QuadOut qo[2];
        for (int tt = 0; tt < 2; ++tt) qo[tt].vo = icsfft::quad_lane(a, u, mem.lout, t, tt, qo[tt].rows, qo[tt].X);
        const bool edge = u.ox[0] < a.ox0 || u.ox[0] + a.V > a.ox1 || u.ox[1] < a.ox0 || u.ox[1] + a.V > a.ox1;    // (as the kernel)
        for (int i = 0; i < 4; ++i) {
          v4f r[2];
          icsfft::read_quads(lds.data(), t, i, r);
          if (mode == 0) { r[0] -= fimg[0][i]; r[1] -= fimg[1][i]; }
          else { icsfft::maxima_quad<true>(a, u, t, 0, i, r[0], o, mx, qo[0], edge); icsfft::maxima_quad<true>(a, u, t, 1, i, r[1], o, mx, qo[1], edge); }
          icsfft::store_quad_at(a, mem, qo[0], edge, i, r[0]); icsfft::store_quad_at(a, mem, qo[1], edge, i, r[1]);
        }
      }
    }
    if (g_planar) from_planar(h, pout, out);
    double wa;
    const double rel = check(h, mode, out, 1, &wa);
    printf("emulation %d x %d, K = %d, mode %d: max |d| = %.3e, relative to max |conv| = %.3e  %s\n", M, N, K, mode, wa, rel, rel < 5e-6 ? "OK" : "FAIL");
    if (!(rel < 5e-6)) rc = 1;
  }
  return rc;
}

// float64 direct sum of one tap of the PSF gradient: gradk[a][b][c] = sum over the interior of e'[y][x] u[y + pad - a][x + pad - b]
double direct_gradk(const Host& h, const std::vector<float>& e, int a, int b, int c) {
  const IcsGeom& g = h.g;
  double s = 0.0;
  for (int y = g.pad; y < g.pad + g.M; ++y)
    for (int x = g.pad; x < g.pad + g.N; ++x)
      s += (double)e[h.org + (size_t)y * g.pitch + 3 * x + c] * (double)h.u[h.org + (size_t)(y + g.pad - a) * g.pitch + 3 * (x + g.pad - b) + c];
  return s;
}
double check_gradk(const Host& h, const std::vector<float>& e, const std::vector<float>& gk, int stride) {
  const int K = h.g.K;
  double worst = 0, m = 0;
  for (int a = 0; a < K; a += stride)
    for (int b = 0; b < K; b += (a % 2 ? stride : 1))
      for (int c = 0; c < 3; ++c) {
        const double r = direct_gradk(h, e, a, b, c);
        m = fmax(m, fabs(r)); worst = fmax(worst, fabs(r - (double)gk[((size_t)a * K + b) * 3 + c]));
      }
  return worst / m;
}

int emulate_gradk(int M, int K, int N) {
  Host h = make_host(M, N, K);
  // a residual with structure (as in the loop: e' = conv(u) - image), not the harness' white noise alone
  for (size_t i = 0; i < h.nf; ++i) h.e[i] = h.e[i] + 0.01f * (h.u[i] - 0.5f) * (h.f[i] != 0.f);
  to_planar(h, h.e, h.pe);
  std::vector<v2f> lds((size_t)ICS_FFT_P * ICS_FFT_PITCH + 128), twl(ICS_FFT_TW_ENTRIES);
  for (int t = 0; t < ICS_FFT_TW_ENTRIES; ++t) twl[t] = icsfft::tw128((t / ICS_FFT_TWS) * (t % ICS_FFT_TWS));
  IcsConvArgs c = conv_args(h, 0, h.pu.data(), h.pe.data(), h.pe.data(), h.pu.data(), h.pu.data(), nullptr);
  IcsFftArgs a;
  ics_conv_fft_fill_args(0, c, nullptr, &a);
  a.planar = 63;
  const icsfft::Mem mem = icsfft::make_mem(a);
  std::vector<float> gk((size_t)K * K * 3, 0.f);
  const int npairs = (a.ntiles + 1) / 2;
  for (int ch = 0; ch < 3; ++ch) {           // one "workgroup" per channel
    std::vector<v2f> acc((size_t)1024 * 16, (v2f){0.f, 0.f});
    for (int p = 0; p < npairs; ++p) {
      const icsfft::Unit u = icsfft::decode_unit(a, 3 * p + ch);
      std::vector<v2f> ze((size_t)1024 * 16), zu((size_t)1024 * 16);
      for (int pass = 0; pass < 2; ++pass) {
        for (int t = 0; t < 1024; ++t) { v4f q[2][4]; if (pass == 0) icsfft::load_image(a, mem, u, t, q); else icsfft::load_window(a, mem, u, t, q); icsfft::store_window(q, lds.data(), t); }
        for (int t = 0; t < 1024; ++t) icsfft::stage_a(lds.data(), t);
        for (int t = 0; t < 1024; ++t) icsfft::stage_b<1>(lds.data(), t);
        { const std::vector<v2f> snap = lds; for (int t = 0; t < 1024; ++t) icsfft::stage_c(snap.data(), lds.data(), twl.data(), t); }
        for (int t = 0; t < 1024; ++t) { v2f z[2][8]; icsfft::stage_d_forward(lds.data(), t, z); for (int i = 0; i < 16; ++i) (pass ? zu : ze)[(size_t)t * 16 + i] = z[i >> 3][i & 7]; }
      }
      for (size_t i = 0; i < acc.size(); ++i) acc[i] += icsfft::cmulc(zu[i], ze[i]);
    }
    for (int t = 0; t < 1024; ++t) { v2f z[2][8]; for (int i = 0; i < 16; ++i) z[i >> 3][i & 7] = acc[(size_t)t * 16 + i]; icsfft::stage_d_inverse(z, lds.data(), t); }
    { const std::vector<v2f> snap = lds; for (int t = 0; t < 1024; ++t) icsfft::stage_e(snap.data(), lds.data(), twl.data(), t); }
    for (int t = 0; t < 1024; ++t) icsfft::stage_b<-1>(lds.data(), t);
    for (int t = 0; t < 1024; ++t) icsfft::stage_g(lds.data(), t);
    for (int aa = 0; aa < K; ++aa)
      for (int bb = 0; bb < K; ++bb) gk[((size_t)aa * K + bb) * 3 + ch] = lds[(K - 1 - aa) * ICS_FFT_PITCH + (K - 1 - bb)].x / 16384.f;
  }
  const double rel = check_gradk(h, h.e, gk, 3);
  printf("emulation %d x %d, K = %d, PSF gradient: relative to max |gradk| = %.3e  %s\n", M, N, K, rel, rel < 1e-5 ? "OK" : "FAIL");
  return rel < 1e-5 ? 0 : 1;
}


// the fused A11 + A13 unit (k_synth_gradk_fft), stage by stage as the kernel runs them: e' (every tile stored) against float64 direct sums and
// BIT FOR BIT against the emulation of k_conv_fft<0>; the K x K x 3 gradient against float64 sums over that e'
int emulate_fused(int M, int K, int N) {
  Host h = make_host(M, N, K);
  std::vector<v2f> spec;
  host_spectrum(h, 0, spec);
  std::vector<v2f> lds((size_t)ICS_FFT_P * ICS_FFT_PITCH + 128), twl(ICS_FFT_TW_ENTRIES);
  for (int t = 0; t < ICS_FFT_TW_ENTRIES; ++t) twl[t] = icsfft::tw128((t / ICS_FFT_TWS) * (t % ICS_FFT_TWS));
  std::vector<float> pout(h.pnf, 0.f);
  IcsFftArgs a;
  ics_conv_fft_fill_args(0, conv_args(h, 0, h.pu.data(), pout.data(), h.pf.data(), h.pu.data(), h.pu.data(), nullptr), (const float*)spec.data(), &a);
  a.planar = 63; a.store_all = 1;
  const icsfft::Mem mem = icsfft::make_mem(a, 0);
  std::vector<float> gk((size_t)K * K * 3, 0.f);
  const int npairs = (a.ntiles + 1) / 2;
  for (int ch = 0; ch < 3; ++ch) {
    std::vector<v2f> acc((size_t)1024 * 16, (v2f){0.f, 0.f}), zu((size_t)1024 * 16);
    for (int p = 0; p < npairs; ++p) {
      const icsfft::Unit u = icsfft::decode_unit(a, 3 * p + ch);
      for (int t = 0; t < 1024; ++t) { v4f pw[2][4]; icsfft::load_window(a, mem, u, t, pw); icsfft::store_window(pw, lds.data(), t); }
      for (int t = 0; t < 1024; ++t) icsfft::stage_a(lds.data(), t);
      for (int t = 0; t < 1024; ++t) icsfft::stage_b<1>(lds.data(), t);
      { const std::vector<v2f> snap = lds; for (int t = 0; t < 1024; ++t) icsfft::stage_c<4>(snap.data(), lds.data(), twl.data(), t); }
      for (int t = 0; t < 1024; ++t) {
        v2f sp[2][8], z[2][8];
        icsfft::load_spectrum(mem, u.c, t, sp);
        icsfft::stage_d_keep(sp, lds.data(), t, z);
        for (int i = 0; i < 16; ++i) zu[(size_t)t * 16 + i] = z[i >> 3][i & 7];
      }
      { const std::vector<v2f> snap = lds; for (int t = 0; t < 1024; ++t) icsfft::stage_e_lean(snap.data(), lds.data(), twl.data(), t); }
      for (int t = 0; t < 1024; ++t) icsfft::stage_b<-1>(lds.data(), t);
      for (int t = 0; t < 1024; ++t) icsfft::stage_g(lds.data(), t);
      for (int t = 0; t < 1024; ++t) {
        v4f fimg[2][4];
        icsfft::load_image(a, mem, u, t, fimg);
        icsfft::QuadOut qo[2];
        for (int tt = 0; tt < 2; ++tt) qo[tt].vo = icsfft::quad_lane(a, u, mem.lout, t, tt, qo[tt].rows, qo[tt].X);
        const bool edge = u.ox[0] < a.ox0 || u.ox[0] + a.V > a.ox1 || u.ox[1] < a.ox0 || u.ox[1] + a.V > a.ox1;
        for (int i = 0; i < 4; ++i) icsfft::residual_quads(a, mem, qo, edge, true, lds.data(), t, i, fimg);
      }
      for (int t = 0; t < 1024; ++t) icsfft::stage_a(lds.data(), t);
      for (int t = 0; t < 1024; ++t) icsfft::stage_b<1>(lds.data(), t);
      { const std::vector<v2f> snap = lds; for (int t = 0; t < 1024; ++t) icsfft::stage_c<4>(snap.data(), lds.data(), twl.data(), t); }
      for (int t = 0; t < 1024; ++t) {
        v2f z[2][8], ac[2][8];
        for (int i = 0; i < 16; ++i) { z[i >> 3][i & 7] = zu[(size_t)t * 16 + i]; ac[i >> 3][i & 7] = acc[(size_t)t * 16 + i]; }
        icsfft::stage_d_acc(lds.data(), t, z, ac);
        for (int i = 0; i < 16; ++i) acc[(size_t)t * 16 + i] = ac[i >> 3][i & 7];
      }
    }
    for (int t = 0; t < 1024; ++t) { v2f z[2][8]; for (int i = 0; i < 16; ++i) z[i >> 3][i & 7] = acc[(size_t)t * 16 + i]; icsfft::stage_d_inverse(z, lds.data(), t); }
    { const std::vector<v2f> snap = lds; for (int t = 0; t < 1024; ++t) icsfft::stage_e(snap.data(), lds.data(), twl.data(), t); }
    for (int t = 0; t < 1024; ++t) icsfft::stage_b<-1>(lds.data(), t);
    for (int t = 0; t < 1024; ++t) icsfft::stage_g(lds.data(), t);
    for (int aa = 0; aa < K; ++aa)
      for (int bb = 0; bb < K; ++bb) gk[((size_t)aa * K + bb) * 3 + ch] = lds[(K - 1 - aa) * ICS_FFT_PITCH + (K - 1 - bb)].x / 16384.f;
  }
  std::vector<float> e(h.nf, 0.f);
  from_planar(h, pout, e);
  double wa;
  const double rel = check(h, 0, e, 1, &wa);
  // everything of the residual frame outside the interior must still be zero (the gradient sums over the interior only; check() looked inside)
  const double relg = check_gradk(h, e, gk, 3);
  printf("emulation %d x %d, K = %d, fused A11 + A13: residual relative to max |conv| = %.3e, PSF gradient relative to max |gradk| = %.3e  %s\n", M, N, K, rel, relg,
         (rel < 5e-6 && relg < 1e-5) ? "OK" : "FAIL");
  return (rel < 5e-6 && relg < 1e-5) ? 0 : 1;
}

// float64 reference of A1 + A2 + A3 from u and the image alone: e = conv(u) - image on the interior (double), g = corr(e zero-extended)
void reference_conv2(const Host& h, std::vector<double>& gref) {
  const IcsGeom& g = h.g;
  std::vector<double> e((size_t)g.uM * g.uN * 3, 0.0);
  for (int y = g.pad; y < g.pad + g.M; ++y)
    for (int x = g.pad; x < g.pad + g.N; ++x)
      for (int c = 0; c < 3; ++c) e[((size_t)y * g.uN + x) * 3 + c] = direct(h, 0, y, x, c) - (double)h.f[h.org + (size_t)y * g.pitch + 3 * x + c];
  gref.assign((size_t)g.uM * g.uN * 3, 0.0);
  for (int y = 0; y < g.uM; ++y)
    for (int x = 0; x < g.uN; ++x)
      for (int c = 0; c < 3; ++c) {
        double s = 0.0;
        for (int a = 0; a < g.K; ++a) {
          const int yy = y + a - g.pad;
          if (yy < 0 || yy >= g.uM) continue;
          for (int b = 0; b < g.K; ++b) {
            const int xx = x + b - g.pad;
            if (xx < 0 || xx >= g.uN) continue;
            s += (double)h.psf[((size_t)a * g.K + b) * 3 + c] * e[((size_t)yy * g.uN + xx) * 3 + c];
          }
        }
        gref[((size_t)y * g.uN + x) * 3 + c] = s;
      }
}
double check_conv2(const Host& h, const std::vector<float>& out, const std::vector<double>& gref, double* worst_abs) {
  const IcsGeom& g = h.g;
  double worst = 0, m = 0;
  for (int y = 0; y < g.uM; ++y)
    for (int x = 0; x < g.uN; ++x)
      for (int c = 0; c < 3; ++c) {
        const double r = gref[((size_t)y * g.uN + x) * 3 + c];
        m = fmax(m, fabs(r));
        const double d = fabs(r - (double)out[h.org + (size_t)y * g.pitch + 3 * x + c]);
        if (getenv("ICS_FFT_DEBUG") && d > 1e-4) fprintf(stderr, "conv2 %d %d %d got %.6g want %.6g\n", y, x, c, out[h.org + (size_t)y * g.pitch + 3 * x + c], r);
        worst = fmax(worst, d);
      }
  *worst_abs = worst;
  return worst / m;
}

// the stage sequence of k_conv_fft<mode> (mode 0 / 1) on the host: `in` -> `pout` (planar)
void emulate_mode(const Host& h, int mode, const std::vector<float>& pin, std::vector<float>& pout) {
  std::vector<v2f> lds((size_t)ICS_FFT_P * ICS_FFT_PITCH + 128), twl(ICS_FFT_TW_ENTRIES), spec;
  for (int t = 0; t < ICS_FFT_TW_ENTRIES; ++t) twl[t] = icsfft::tw128((t / ICS_FFT_TWS) * (t % ICS_FFT_TWS));
  host_spectrum(h, mode, spec);
  pout.assign(h.pnf, 0.f);
  uint32_t red[16] = {0};
  IcsFftArgs a;
  ics_conv_fft_fill_args(mode, conv_args(h, mode, pin.data(), pout.data(), h.pf.data(), h.pu.data(), h.put.data(), red), (const float*)spec.data(), &a);
  a.planar = 63;
  const icsfft::Mem mem = icsfft::make_mem(a);
  for (int n = 0; n < a.nunits; ++n) {
    const icsfft::Unit u = icsfft::decode_unit(a, n);
    for (int t = 0; t < 1024; ++t) { v4f pw[2][4]; icsfft::load_window(a, mem, u, t, pw); icsfft::store_window(pw, lds.data(), t); }
    for (int t = 0; t < 1024; ++t) icsfft::stage_a(lds.data(), t);
    for (int t = 0; t < 1024; ++t) icsfft::stage_b<1>(lds.data(), t);
    { const std::vector<v2f> snap = lds; for (int t = 0; t < 1024; ++t) icsfft::stage_c(snap.data(), lds.data(), twl.data(), t); }
    for (int t = 0; t < 1024; ++t) { v2f sp[2][8]; icsfft::load_spectrum(mem, u.c, t, sp); icsfft::stage_d(sp, lds.data(), t); }
    { const std::vector<v2f> snap = lds; for (int t = 0; t < 1024; ++t) icsfft::stage_e(snap.data(), lds.data(), twl.data(), t); }
    for (int t = 0; t < 1024; ++t) icsfft::stage_b<-1>(lds.data(), t);
    for (int t = 0; t < 1024; ++t) icsfft::stage_g(lds.data(), t);
    for (int t = 0; t < 1024; ++t) {
      icsfft::Maxima mx; icsfft::maxima_init(mx);
      v4f fimg[2][4];
      icsfft::Ops o;
      if (mode == 0) icsfft::load_image(a, mem, u, t, fimg);
      else { icsfft::load_ops<false>(a, mem, u, t, 0, o); icsfft::load_ops<false>(a, mem, u, t, 1, o); }
      icsfft::QuadOut qo[2];
      for (int tt = 0; tt < 2; ++tt) qo[tt].vo = icsfft::quad_lane(a, u, mem.lout, t, tt, qo[tt].rows, qo[tt].X);
      const bool edge = u.ox[0] < a.ox0 || u.ox[0] + a.V > a.ox1 || u.ox[1] < a.ox0 || u.ox[1] + a.V > a.ox1;
      for (int i = 0; i < 4; ++i) {
        v4f r[2];
        icsfft::read_quads(lds.data(), t, i, r);
        if (mode == 0) { r[0] -= fimg[0][i]; r[1] -= fimg[1][i]; }
        else { icsfft::maxima_quad<false>(a, u, t, 0, i, r[0], o, mx, qo[0], edge); icsfft::maxima_quad<false>(a, u, t, 1, i, r[1], o, mx, qo[1], edge); }
        icsfft::store_quad_at(a, mem, qo[0], edge, i, r[0]); icsfft::store_quad_at(a, mem, qo[1], edge, i, r[1]);
      }
    }
  }
}

// mode 2 (k_conv_fft<2>), stage by stage as the kernel runs them, with the image spectra of k_fft_image_spectrum
int emulate_conv2(int M, int K, int N) {
  Host h = make_host(M, N, K);
  // a realistic residual: the image is the frame's own synthesis plus noise (make_host's image is unrelated to u: |e| ~ |u|, which hides
  // cancellation in the frequency domain)
  for (int y = h.g.pad; y < h.g.pad + M; ++y)
    for (int x = h.g.pad; x < h.g.pad + N; ++x)
      for (int c = 0; c < 3; ++c) {
        const size_t o = h.org + (size_t)y * h.g.pitch + 3 * x + c;
        h.f[o] = (float)(direct(h, 0, y, x, c) + 5e-3 * ((double)rand() / RAND_MAX - 0.5));
      }
  to_planar(h, h.f, h.pf);
  std::vector<v2f> spec0, spec1;
  host_spectrum(h, 0, spec0); host_spectrum(h, 1, spec1);
  std::vector<v2f> lds((size_t)ICS_FFT_P * ICS_FFT_PITCH + 128), twl(ICS_FFT_TW_ENTRIES);
  for (int t = 0; t < ICS_FFT_TW_ENTRIES; ++t) twl[t] = icsfft::tw128((t / ICS_FFT_TWS) * (t % ICS_FFT_TWS));
  std::vector<float> pout(h.pnf, 0.f);
  uint32_t red[16] = {0};
  IcsFftArgs a;
  ics_conv_fft_fill_args(2, conv_args(h, 1, h.pu.data(), pout.data(), h.pf.data(), h.pu.data(), h.put.data(), red), (const float*)spec0.data(), &a);
  a.planar = 63; a.spec1 = spec1.data();
  std::vector<float> fspec((size_t)a.nunits * 8 * 1024 * 4, 0.f);
  a.fspec = fspec.data();
  printf("mode 2: V %d x %d, tiles %d (x %d), units %d\n", a.Vy, a.V, a.ntiles, a.tiles_x, a.nunits);
  {   // k_fft_image_spectrum
    IcsFftArgs b = a;
    b.c.in = b.c.f; b.wpad = h.g.pad;
    const icsfft::Mem mem = icsfft::make_mem(b, 2);
    for (int n = 0; n < b.nunits; ++n) {
      const icsfft::Unit u = icsfft::decode_unit(b, n);
      for (int t = 0; t < 1024; ++t) { v4f pw[2][4]; icsfft::load_window(b, mem, u, t, pw); icsfft::store_window(pw, lds.data(), t); }
      for (int t = 0; t < 1024; ++t) icsfft::stage_a(lds.data(), t);
      for (int t = 0; t < 1024; ++t) icsfft::stage_b<1>(lds.data(), t);
      { const std::vector<v2f> snap = lds; for (int t = 0; t < 1024; ++t) icsfft::stage_c(snap.data(), lds.data(), twl.data(), t); }
      for (int t = 0; t < 1024; ++t) {
        v2f z[2][8];
        icsfft::stage_d_forward(lds.data(), t, z);
        icsfft::store_spectrum(mem.fspec, 8 * n, t, z);
      }
    }
  }
  const icsfft::Mem mem = icsfft::make_mem(a, 2);
  int nborder = 0;
  for (int n = 0; n < a.nunits; ++n) {
    const icsfft::Unit u = icsfft::decode_unit(a, n);
    for (int t = 0; t < 1024; ++t) { v4f pw[2][4]; icsfft::load_window(a, mem, u, t, pw); icsfft::store_window(pw, lds.data(), t); }
    for (int t = 0; t < 1024; ++t) icsfft::stage_a(lds.data(), t);
    for (int t = 0; t < 1024; ++t) icsfft::stage_b<1>(lds.data(), t);
    if (!icsfft::unit_is_border(a, u)) {
      { const std::vector<v2f> snap = lds; for (int t = 0; t < 1024; ++t) icsfft::stage_c<4>(snap.data(), lds.data(), twl.data(), t); }
      for (int t = 0; t < 1024; ++t)
        for (int hf = 0; hf < 2; ++hf) {
          v2f fs[8], s0[8], s1[8];
          icsfft::load_spectrum_half<1>(mem.fspec, 8 * n, t, hf, fs); icsfft::load_spectrum_half<1>(mem.spec, 8 * u.c, t, hf, s0); icsfft::load_spectrum_half<1>(mem.spec1, 8 * u.c, t, hf, s1);
          icsfft::stage_d2_half(s0, s1, fs, lds.data(), t, hf);
        }
    } else {
      ++nborder;
      { const std::vector<v2f> snap = lds; for (int t = 0; t < 1024; ++t) icsfft::stage_c(snap.data(), lds.data(), twl.data(), t); }
      for (int t = 0; t < 1024; ++t) { v2f sp[2][8]; icsfft::load_spectrum(mem, u.c, t, sp); icsfft::stage_d(sp, lds.data(), t); }
      { const std::vector<v2f> snap = lds; for (int t = 0; t < 1024; ++t) icsfft::stage_e(snap.data(), lds.data(), twl.data(), t); }
      for (int t = 0; t < 1024; ++t) icsfft::stage_b<-1>(lds.data(), t);
      for (int t = 0; t < 1024; ++t) icsfft::stage_g(lds.data(), t);
      for (int t = 0; t < 1024; ++t) icsfft::residual_window(a, mem, u, lds.data(), t);
      for (int t = 0; t < 1024; ++t) icsfft::stage_a(lds.data(), t);
      for (int t = 0; t < 1024; ++t) icsfft::stage_b<1>(lds.data(), t);
      { const std::vector<v2f> snap = lds; for (int t = 0; t < 1024; ++t) icsfft::stage_c(snap.data(), lds.data(), twl.data(), t); }
      for (int t = 0; t < 1024; ++t) {
        v2f sp[2][8];
        for (int hf = 0; hf < 2; ++hf) icsfft::load_spectrum_half<1>(mem.spec1, 8 * u.c, t, hf, sp[hf]);
        icsfft::stage_d(sp, lds.data(), t);
      }
    }
    { const std::vector<v2f> snap = lds; for (int t = 0; t < 1024; ++t) icsfft::stage_e(snap.data(), lds.data(), twl.data(), t); }
    for (int t = 0; t < 1024; ++t) icsfft::stage_b<-1>(lds.data(), t);
    for (int t = 0; t < 1024; ++t) icsfft::stage_g(lds.data(), t);
    for (int t = 0; t < 1024; ++t) {
      icsfft::Maxima mx; icsfft::maxima_init(mx);
      icsfft::Ops o;
      icsfft::load_ops<false>(a, mem, u, t, 0, o); icsfft::load_ops<false>(a, mem, u, t, 1, o);
      icsfft::QuadOut qo[2];
      for (int tt = 0; tt < 2; ++tt) qo[tt].vo = icsfft::quad_lane(a, u, mem.lout, t, tt, qo[tt].rows, qo[tt].X);
      const bool edge = u.ox[0] < a.ox0 || u.ox[0] + a.V > a.ox1 || u.ox[1] < a.ox0 || u.ox[1] + a.V > a.ox1;
      for (int i = 0; i < 4; ++i) {
        v4f r[2];
        icsfft::read_quads(lds.data(), t, i, r);
        icsfft::maxima_quad<false>(a, u, t, 0, i, r[0], o, mx, qo[0], edge); icsfft::maxima_quad<false>(a, u, t, 1, i, r[1], o, mx, qo[1], edge);
        icsfft::store_quad_at(a, mem, qo[0], edge, i, r[0]); icsfft::store_quad_at(a, mem, qo[1], edge, i, r[1]);
      }
    }
  }
  std::vector<float> out(h.nf, 0.f);
  from_planar(h, pout, out);
  std::vector<double> gref;
  reference_conv2(h, gref);
  double wa;
  const double rel = check_conv2(h, out, gref, &wa);
  // the yardstick: the two kernels mode 2 replaces, on the same frame against the same float64 reference.  The back-projection of a small
  // residual e = conv(u) - image inherits conv's absolute rounding (~1e-7 |u|) whatever the path: errors are quoted against max |gradu| and
  // held to twice what the two-kernel path shows (+ 1e-6)
  std::vector<float> pe, pg, g2(h.nf, 0.f);
  emulate_mode(h, 0, h.pu, pe);
  emulate_mode(h, 1, pe, pg);
  from_planar(h, pg, g2);
  double wa2;
  const double rel2 = check_conv2(h, g2, gref, &wa2);
  const bool ok = rel < 4 * rel2 + 1e-6 && wa < 2.5e-6;      // (the second bound: 5e-6 of max |conv(u)| ~ 0.5, the convolutions' own stage gate)
  printf("emulation %d x %d, K = %d, mode 2 (A1 + A3 in one unit, %d of %d units on the outer ring): max |d| = %.3e, relative to max |gradu| = %.3e (the two kernels: %.3e)  %s\n", M, N, K, nborder,
         a.nunits, wa, rel, rel2, ok ? "OK" : "FAIL");
  return ok ? 0 : 1;
}

// spectra of the tap blocks on the host: block q = qa * nb + qb of Kb x Kb taps of orientation o, taps beyond K zero (as k_fft_spectrum with blocks)
void host_spectrum_blk(const Host& h, int o, int nb, int Kb, std::vector<v2f>& spec) {
  const int K = h.g.K;
  spec.assign((size_t)nb * nb * 3 * 128 * 128, (v2f){0.f, 0.f});
  std::vector<double> G((size_t)Kb * 128 * 2);
  for (int q = 0; q < nb * nb; ++q) {
    const int a0 = (q / nb) * Kb, b0 = (q % nb) * Kb;
    for (int c = 0; c < 3; ++c) {
      for (int a = 0; a < Kb; ++a)
        for (int kx = 0; kx < 128; ++kx) {
          double re = 0, im = 0;
          for (int b = 0; b < Kb; ++b) {
            const int ta = a0 + a, tb = b0 + b;
            const double w = (ta < K && tb < K) ? (o == 0 ? h.psf[((size_t)(K - 1 - ta) * K + (K - 1 - tb)) * 3 + c] : h.psf[((size_t)ta * K + tb) * 3 + c]) : 0.0;
            const double ph = -2.0 * M_PI * ((b * kx) & 127) / 128.0;
            re += w * cos(ph); im += w * sin(ph);
          }
          G[((size_t)a * 128 + kx) * 2] = re; G[((size_t)a * 128 + kx) * 2 + 1] = im;
        }
      for (int ky = 0; ky < 128; ++ky)
        for (int kx = 0; kx < 128; ++kx) {
          double re = 0, im = 0;
          for (int a = 0; a < Kb; ++a) {
            const double ph = -2.0 * M_PI * ((a * ky) & 127) / 128.0, wr = cos(ph), wi = sin(ph);
            const double gr = G[((size_t)a * 128 + kx) * 2], gi = G[((size_t)a * 128 + kx) * 2 + 1];
            re += gr * wr - gi * wi; im += gr * wi + gi * wr;
          }
          spec[(size_t)q * 3 * 128 * 128 + icsfft::spec_index(c, ky, kx)] = (v2f){(float)(re / 16384.0), (float)(-im / 16384.0)};
        }
    }
  }
}

// PSF sizes above the single-tile range: k_conv_fft_blk and the lag blocks of k_gradk_fft, stage by stage
int emulate_blk(int M, int K, int N) {
  Host h = make_host(M, N, K);
  int nb, Kb;
  ics_conv_fft_blk_shape(K, &nb, &Kb);
  std::vector<v2f> lds((size_t)ICS_FFT_P * ICS_FFT_PITCH + 128), twl(ICS_FFT_TW_ENTRIES);
  for (int t = 0; t < ICS_FFT_TW_ENTRIES; ++t) twl[t] = icsfft::tw128((t / ICS_FFT_TWS) * (t % ICS_FFT_TWS));
  int rc = 0;
  for (int mode = 0; mode < 2; ++mode) {
    std::vector<v2f> spec;
    host_spectrum_blk(h, mode, nb, Kb, spec);
    std::vector<float> out(h.nf, 0.f), pout(h.pnf, 0.f);
    uint32_t red[16] = {0};
    IcsFftArgs a;
    ics_conv_fft_fill_args(mode, conv_args(h, mode, mode == 0 ? h.pu.data() : h.pe.data(), pout.data(), h.pf.data(), h.pu.data(), h.put.data(), red), (const float*)spec.data(), &a, nb, Kb);
    a.planar = 63;
    const icsfft::Mem mem = icsfft::make_mem(a);
    for (int n = 0; n < a.nunits; ++n) {
      const icsfft::Unit u = icsfft::decode_unit(a, n);
      std::vector<v2f> acc((size_t)1024 * 16, (v2f){0.f, 0.f});
      for (int b = 0; b < nb * nb; ++b) {
        for (int t = 0; t < 1024; ++t) { v4f pw[2][4]; icsfft::load_window(a, mem, u, t, pw, 0, 2, (b / nb) * Kb, (b % nb) * Kb); icsfft::store_window(pw, lds.data(), t); }
        for (int t = 0; t < 1024; ++t) icsfft::stage_a(lds.data(), t);
        for (int t = 0; t < 1024; ++t) icsfft::stage_b<1>(lds.data(), t);
        { const std::vector<v2f> snap = lds; for (int t = 0; t < 1024; ++t) icsfft::stage_c<4>(snap.data(), lds.data(), twl.data(), t); }
        for (int t = 0; t < 1024; ++t) {
          v2f sp[2][8], ac[2][8];
          for (int hf = 0; hf < 2; ++hf) icsfft::load_spectrum_half<1>(mem.spec, 8 * (3 * b + u.c), t, hf, sp[hf]);
          for (int i = 0; i < 16; ++i) ac[i >> 3][i & 7] = acc[(size_t)t * 16 + i];
          icsfft::stage_d_mac(lds.data(), t, sp, ac);
          for (int i = 0; i < 16; ++i) acc[(size_t)t * 16 + i] = ac[i >> 3][i & 7];
        }
      }
      for (int t = 0; t < 1024; ++t) { v2f z[2][8]; for (int i = 0; i < 16; ++i) z[i >> 3][i & 7] = acc[(size_t)t * 16 + i]; icsfft::stage_d_inverse(z, lds.data(), t); }
      { const std::vector<v2f> snap = lds; for (int t = 0; t < 1024; ++t) icsfft::stage_e(snap.data(), lds.data(), twl.data(), t); }
      for (int t = 0; t < 1024; ++t) icsfft::stage_b<-1>(lds.data(), t);
      for (int t = 0; t < 1024; ++t) icsfft::stage_g(lds.data(), t);
      for (int t = 0; t < 1024; ++t) {
        icsfft::Maxima mx; icsfft::maxima_init(mx);
        v4f fimg[2][4];
        icsfft::Ops o;
        if (mode == 0) icsfft::load_image(a, mem, u, t, fimg);
        else { icsfft::load_ops<false>(a, mem, u, t, 0, o); icsfft::load_ops<false>(a, mem, u, t, 1, o); }
        icsfft::QuadOut qo[2];
        for (int tt = 0; tt < 2; ++tt) qo[tt].vo = icsfft::quad_lane(a, u, mem.lout, t, tt, qo[tt].rows, qo[tt].X);
        const bool edge = u.ox[0] < a.ox0 || u.ox[0] + a.V > a.ox1 || u.ox[1] < a.ox0 || u.ox[1] + a.V > a.ox1;
        for (int i = 0; i < 4; ++i) {
          v4f r[2];
          icsfft::read_quads(lds.data(), t, i, r);
          if (mode == 0) { r[0] -= fimg[0][i]; r[1] -= fimg[1][i]; }
          else { icsfft::maxima_quad<false>(a, u, t, 0, i, r[0], o, mx, qo[0], edge); icsfft::maxima_quad<false>(a, u, t, 1, i, r[1], o, mx, qo[1], edge); }
          icsfft::store_quad_at(a, mem, qo[0], edge, i, r[0]); icsfft::store_quad_at(a, mem, qo[1], edge, i, r[1]);
        }
      }
    }
    from_planar(h, pout, out);
    double wa;
    const double rel = check(h, mode, out, 1, &wa);
    printf("emulation %d x %d, K = %d as %d x %d tap blocks of %d, mode %d (%d units of %d x %d valid): max |d| = %.3e, relative to max |conv| = %.3e  %s\n", M, N, K, nb, nb, Kb, mode, a.nunits, a.Vy, a.V, wa, rel,
           rel < 5e-6 ? "OK" : "FAIL");
    if (!(rel < 5e-6)) rc = 1;
  }
  {   // the gradient's lag blocks
    for (size_t i = 0; i < h.nf; ++i) h.e[i] = h.e[i] + 0.01f * (h.u[i] - 0.5f) * (h.f[i] != 0.f);
    to_planar(h, h.e, h.pe);
    IcsConvArgs c = conv_args(h, 0, h.pu.data(), h.pe.data(), h.pe.data(), h.pu.data(), h.pu.data(), nullptr);
    std::vector<float> gk((size_t)K * K * 3, 0.f);
    for (int qy = 0; qy < nb; ++qy)
      for (int qx = 0; qx < nb; ++qx) {
        IcsFftArgs a;
        ics_conv_fft_fill_args(0, c, nullptr, &a, nb, Kb);
        a.planar = 63; a.lag_y = qy * Kb; a.lag_x = qx * Kb;
        const icsfft::Mem mem = icsfft::make_mem(a);
        const int npairs = (a.ntiles + 1) / 2;
        for (int ch = 0; ch < 3; ++ch) {
          std::vector<v2f> acc((size_t)1024 * 16, (v2f){0.f, 0.f});
          for (int p = 0; p < npairs; ++p) {
            const icsfft::Unit u = icsfft::decode_unit(a, 3 * p + ch);
            std::vector<v2f> ze((size_t)1024 * 16), zu((size_t)1024 * 16);
            for (int pass = 0; pass < 2; ++pass) {
              for (int t = 0; t < 1024; ++t) { v4f q[2][4]; if (pass == 0) icsfft::load_image(a, mem, u, t, q); else icsfft::load_window(a, mem, u, t, q, 0, 2, a.lag_y, a.lag_x); icsfft::store_window(q, lds.data(), t); }
              for (int t = 0; t < 1024; ++t) icsfft::stage_a(lds.data(), t);
              for (int t = 0; t < 1024; ++t) icsfft::stage_b<1>(lds.data(), t);
              { const std::vector<v2f> snap = lds; for (int t = 0; t < 1024; ++t) icsfft::stage_c(snap.data(), lds.data(), twl.data(), t); }
              for (int t = 0; t < 1024; ++t) { v2f z[2][8]; icsfft::stage_d_forward(lds.data(), t, z); for (int i = 0; i < 16; ++i) (pass ? zu : ze)[(size_t)t * 16 + i] = z[i >> 3][i & 7]; }
            }
            for (size_t i = 0; i < acc.size(); ++i) acc[i] += icsfft::cmulc(zu[i], ze[i]);
          }
          for (int t = 0; t < 1024; ++t) { v2f z[2][8]; for (int i = 0; i < 16; ++i) z[i >> 3][i & 7] = acc[(size_t)t * 16 + i]; icsfft::stage_d_inverse(z, lds.data(), t); }
          { const std::vector<v2f> snap = lds; for (int t = 0; t < 1024; ++t) icsfft::stage_e(snap.data(), lds.data(), twl.data(), t); }
          for (int t = 0; t < 1024; ++t) icsfft::stage_b<-1>(lds.data(), t);
          for (int t = 0; t < 1024; ++t) icsfft::stage_g(lds.data(), t);
          for (int ly = 0; ly < Kb; ++ly)
            for (int lx = 0; lx < Kb; ++lx) {
              const int aa = K - 1 - a.lag_y - ly, bb = K - 1 - a.lag_x - lx;
              if (aa >= 0 && bb >= 0) gk[((size_t)aa * K + bb) * 3 + ch] = lds[ly * ICS_FFT_PITCH + lx].x / 16384.f;
            }
        }
      }
    const double rel = check_gradk(h, h.e, gk, 5);
    printf("emulation %d x %d, K = %d, PSF gradient as %d x %d lag blocks: relative to max |gradk| = %.3e  %s\n", M, N, K, nb, nb, rel, rel < 1e-5 ? "OK" : "FAIL");
    if (!(rel < 1e-5)) rc = 1;
  }
  return rc;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

int gpu(int M, int K, int N, int reps) {
  Host h = make_host(M, N, K);
  float *du, *de, *df, *dut, *dout, *dpsf, *dspec0, *dspec1; uint32_t* dred;
  const size_t fb = (g_planar ? h.pnf : h.nf) * 4;
  for (float** p : {&du, &de, &df, &dut, &dout}) CK(hipMalloc(p, fb));
  CK(hipMalloc(&dpsf, h.psf.size() * 4)); CK(hipMalloc(&dred, 1024)); CK(hipMemset(dred, 0, 1024));
  const size_t sf = ics_conv_fft_spectrum_floats();
  CK(hipMalloc(&dspec0, sf * 4)); CK(hipMalloc(&dspec1, sf * 4));
  CK(hipMemcpy(du, (g_planar ? h.pu : h.u).data(), fb, hipMemcpyHostToDevice)); CK(hipMemcpy(de, (g_planar ? h.pe : h.e).data(), fb, hipMemcpyHostToDevice));
  CK(hipMemcpy(df, (g_planar ? h.pf : h.f).data(), fb, hipMemcpyHostToDevice)); CK(hipMemcpy(dut, (g_planar ? h.put : h.ut).data(), fb, hipMemcpyHostToDevice));
  CK(hipMemcpy(dpsf, h.psf.data(), h.psf.size() * 4, hipMemcpyHostToDevice));
  CK(ics_launch_fft_spectrum(dpsf, K, dspec0, dspec1, 0));
  CK(hipDeviceSynchronize());
  {   // spectrum against the host's float64 evaluation
    std::vector<v2f> hs, ds(sf / 2);
    for (int o = 0; o < 2; ++o) {
      host_spectrum(h, o, hs);
      CK(hipMemcpy(ds.data(), o ? dspec1 : dspec0, sf * 4, hipMemcpyDeviceToHost));
      double w = 0, m = 0;
      for (size_t i = 0; i < hs.size(); ++i) { w = fmax(w, fmax(fabs(hs[i].x - ds[i].x), fabs(hs[i].y - ds[i].y))); m = fmax(m, fabs(hs[i].x)); }
      printf("spectrum %d: max |d| = %.3e of %.3e\n", o, w, m);
    }
  }
  int rc = 0;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int mode = 0; mode < 2; ++mode) {
    CK(hipMemset(dout, 0, fb));
    IcsConvArgs a = conv_args(h, mode, mode == 0 ? du : de, dout, df, du, dut, dred);
    IcsFftArgs fa; ics_conv_fft_fill_args(mode, a, mode == 0 ? dspec0 : dspec1, &fa);
    fa.planar = 63;
    CK(ics_launch_conv_fft_args(mode, fa, 0));
    CK(hipDeviceSynchronize());
    std::vector<float> out(h.nf, 0.f);
    if (g_planar) { std::vector<float> po(h.pnf); CK(hipMemcpy(po.data(), dout, fb, hipMemcpyDeviceToHost)); from_planar(h, po, out); }
    else CK(hipMemcpy(out.data(), dout, h.nf * 4, hipMemcpyDeviceToHost));
    double wa;
    const int step = (M <= 600) ? 1 : (M / 24) | 1;
    const double rel = check(h, mode, out, step, &wa);
    printf("GPU %d x %d, K = %d, mode %d: max |d| = %.3e, relative to max |conv| = %.3e  %s\n", M, N, K, mode, wa, rel, rel < 5e-6 ? "OK" : "FAIL");
    if (!(rel < 5e-6)) rc = 1;
    if (mode == 1) {
      uint32_t red[16];
      CK(hipMemcpy(red, dred, 64, hipMemcpyDeviceToHost));
      printf("  maxima keys -> max|g| %.6g %.6g %.6g   max u %.6g %.6g %.6g\n", ics_key2f(red[0]), ics_key2f(red[1]), ics_key2f(red[2]), ics_key2f(red[3]), ics_key2f(red[4]), ics_key2f(red[5]));
    }
    for (int i = 0; i < 5; ++i) CK(ics_launch_conv_fft_args(mode, fa, 0));
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) CK(ics_launch_conv_fft_args(mode, fa, 0));
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
#ifdef ICS_FFT_TRACE
    {
      long long* dtr; const size_t nt = (size_t)256 * 16 * 16 * 10;
      CK(hipMalloc(&dtr, nt * 8)); CK(hipMemset(dtr, 0, nt * 8));
      fa.trace = dtr;
      CK(ics_launch_conv_fft_args(mode, fa, 0)); CK(hipDeviceSynchronize());
      std::vector<long long> tr(nt);
      CK(hipMemcpy(tr.data(), dtr, nt * 8, hipMemcpyDeviceToHost));
      static const char* nm[9] = {"barrier", "B", "C", "D spec", "E + loads", "F", "G", "epilogue", "A of next"};
      // per stamp: when the FIRST and the LAST of the 16 waves pass it, relative to the unit's first stamp (mean over units)
      double first[10] = {0}, last[10] = {0}, own[9] = {0}; int cnt = 0;
      for (int b = 0; b < 256; b += 5)
        for (int r = 1; r < 8; ++r) {   // rounds 1..7 of every fifth workgroup
          const long long* t = &tr[(((size_t)b * 16 + r) * 16) * 10];
          if (!t[9] || !t[0]) continue;
          long long t0 = t[0];
          for (int w = 0; w < 16; ++w) t0 = t[w * 10] < t0 ? t[w * 10] : t0;
          for (int i = 0; i < 10; ++i) {
            long long f = t[i], l = t[i];
            for (int w = 0; w < 16; ++w) { f = t[w * 10 + i] < f ? t[w * 10 + i] : f; l = t[w * 10 + i] > l ? t[w * 10 + i] : l; }
            first[i] += (double)(f - t0); last[i] += (double)(l - t0);
          }
          for (int i = 0; i < 9; ++i) { double s = 0; for (int w = 0; w < 16; ++w) s += (double)(t[w * 10 + i + 1] - t[w * 10 + i]); own[i] += s / 16; }
          ++cnt;
        }
      printf("  phase timeline (shader clocks from the unit's start, mean of %d units; first wave / last wave to pass each mark; mean time a wave spends in the phase):\n", cnt);
      for (int i = 0; i < 9; ++i) printf("    %-10s ends %6.0f / %6.0f   own %6.0f\n", nm[i], first[i + 1] / cnt, last[i + 1] / cnt, own[i] / cnt);
      fa.trace = nullptr; hipFree(dtr);
    }
#endif
    printf("  mode %d: %.4f ms per launch (%d launches), %d units, %.2f us per unit and CU\n", mode, ms / reps, reps, fa.nunits, 1e3 * ms / reps / ((fa.nunits + 255) / 256));
  }
  {   // the PSF gradient on the tiles
    std::vector<float> es = h.e;
    for (size_t i = 0; i < h.nf; ++i) es[i] = es[i] + 0.01f * (h.u[i] - 0.5f) * (h.f[i] != 0.f);
    std::vector<float> pes; to_planar(h, es, pes);
    CK(hipMemcpy(de, pes.data(), fb, hipMemcpyHostToDevice));
    float *dpart, *dgk;
    CK(hipMalloc(&dpart, (size_t)768 * K * K * 4)); CK(hipMalloc(&dgk, (size_t)3 * K * K * 4));
    CK(ics_launch_gradk_fft(du + h.porg, de + h.porg, h.g, dpart, dgk, 0));
    CK(hipDeviceSynchronize());
    std::vector<float> gk((size_t)3 * K * K);
    CK(hipMemcpy(gk.data(), dgk, gk.size() * 4, hipMemcpyDeviceToHost));
    if ((long)M * N <= 1200L * 1200L) {
      const double rel = check_gradk(h, es, gk, M <= 400 ? 2 : 7);
      printf("GPU %d x %d, K = %d, PSF gradient: relative to max |gradk| = %.3e  %s\n", M, N, K, rel, rel < 1e-5 ? "OK" : "FAIL");
      if (!(rel < 1e-5)) rc = 1;
    }
    for (int i = 0; i < 3; ++i) CK(ics_launch_gradk_fft(du + h.porg, de + h.porg, h.g, dpart, dgk, 0));
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) CK(ics_launch_gradk_fft(du + h.porg, de + h.porg, h.g, dpart, dgk, 0));
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float msg; CK(hipEventElapsedTime(&msg, e0, e1));
    printf("  PSF gradient on the tiles: %.4f ms per launch (kernel + reduction)\n", msg / reps);
  }
  {   // the fused A11 + A13 unit against the two kernels it replaces: bit for bit (same stage functions, same order, same walk), then timed
    float *de2, *dpart, *dgk, *dpart2, *dgk2;
    CK(hipMalloc(&de2, fb)); CK(hipMemset(de2, 0, fb)); CK(hipMemset(dout, 0, fb));
    CK(hipMalloc(&dpart, (size_t)768 * K * K * 4)); CK(hipMalloc(&dgk, (size_t)3 * K * K * 4));
    CK(hipMalloc(&dpart2, (size_t)768 * K * K * 4)); CK(hipMalloc(&dgk2, (size_t)3 * K * K * 4));
    IcsConvArgs a = conv_args(h, 0, du, dout, df, du, dut, dred);
    IcsFftArgs fa; ics_conv_fft_fill_args(0, a, dspec0, &fa);
    fa.planar = 63;
    CK(ics_launch_conv_fft_args(0, fa, 0));
    CK(ics_launch_gradk_fft(du + h.porg, dout + h.porg, h.g, dpart, dgk, 0));
    CK(ics_launch_synth_gradk_fft(du + h.porg, df + h.porg, de2 + h.porg, dspec0, h.g, 0, 0, 0, 0, 1, dpart2, dgk2, 0));
    CK(hipDeviceSynchronize());
    std::vector<float> ra(h.pnf), rb(h.pnf), g1((size_t)3 * K * K), g2((size_t)3 * K * K);
    CK(hipMemcpy(ra.data(), dout, fb, hipMemcpyDeviceToHost)); CK(hipMemcpy(rb.data(), de2, fb, hipMemcpyDeviceToHost));
    CK(hipMemcpy(g1.data(), dgk, g1.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(g2.data(), dgk2, g2.size() * 4, hipMemcpyDeviceToHost));
    size_t nde = 0, ndg = 0; double gm = 0, gd = 0;
    for (size_t i = 0; i < h.pnf; ++i) nde += memcmp(&ra[i], &rb[i], 4) != 0;
    for (size_t i = 0; i < g1.size(); ++i) { ndg += memcmp(&g1[i], &g2[i], 4) != 0; gm = fmax(gm, fabs(g1[i])); gd = fmax(gd, fabs(g1[i] - g2[i])); }
    printf("fused A11 + A13 against k_conv_fft<0> + k_gradk_fft: %zu residual values differ, %zu gradient values differ (max |d| %.3e of %.3e)  %s\n", nde, ndg, gd, gm, (nde || ndg) ? "FAIL" : "OK");
    if (nde || ndg) rc = 1;
    // a window in the middle of the frame: only the tiles under it store
    const int wy0 = h.g.pad + M / 2 - 100, wx0 = h.g.pad + N / 2 - 100;
    for (int all = 1; all >= 0; --all) {
      for (int i = 0; i < 3; ++i) CK(ics_launch_synth_gradk_fft(du + h.porg, df + h.porg, de2 + h.porg, dspec0, h.g, wy0, wy0 + 200, wx0, wx0 + 200, all, dpart2, dgk2, 0));
      CK(hipEventRecord(e0, 0));
      for (int i = 0; i < reps; ++i) CK(ics_launch_synth_gradk_fft(du + h.porg, df + h.porg, de2 + h.porg, dspec0, h.g, wy0, wy0 + 200, wx0, wx0 + 200, all, dpart2, dgk2, 0));
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float msf; CK(hipEventElapsedTime(&msf, e0, e1));
      const int npairs = (fa.ntiles + 1) / 2;
      printf("  fused A11 + A13 (%s): %.4f ms per launch (kernel + reduction), %d pairs x 3 on 255 workgroups = %d rounds, %.2f us per unit\n", all ? "every tile stores e'" : "window tiles store e'",
             msf / reps, npairs, (npairs + 84) / 85, 1e3 * msf / reps / ((npairs + 84) / 85));
    }
    CK(hipMemcpy(g2.data(), dgk2, g2.size() * 4, hipMemcpyDeviceToHost));
    ndg = 0;
    for (size_t i = 0; i < g1.size(); ++i) ndg += memcmp(&g1[i], &g2[i], 4) != 0;
    if (ndg) { printf("  windowed store changed the gradient: FAIL\n"); rc = 1; }
  }
  if (ics_conv2_fft_supported(h.g)) {   // mode 2: A1 + A3 in one unit, against the two kernels (and float64 on small frames), then timed
    // the image = the frame's own synthesis + noise (a realistic, small residual)
    CK(hipMemset(dout, 0, fb));
    IcsConvArgs a0 = conv_args(h, 0, du, dout, df, du, dut, dred);
    {   // f := conv(u) + noise, through mode 0 with a zero image, then noise added on the host
      std::vector<float> zero(h.pnf, 0.f);
      CK(hipMemcpy(df, zero.data(), fb, hipMemcpyHostToDevice));
      IcsFftArgs fa; ics_conv_fft_fill_args(0, a0, dspec0, &fa); fa.planar = 63;
      CK(ics_launch_conv_fft_args(0, fa, 0)); CK(hipDeviceSynchronize());
      std::vector<float> pc(h.pnf);
      CK(hipMemcpy(pc.data(), dout, fb, hipMemcpyDeviceToHost));
      srand(11);
      const IcsGeom& g = h.g; const int pp = ics_ppitch(g); const size_t pl = ics_plane_floats(g);
      for (int c = 0; c < 3; ++c)
        for (int y = g.pad; y < g.pad + M; ++y)
          for (int x = g.pad; x < g.pad + N; ++x) pc[c * pl + h.porg + (size_t)y * pp + x] += 5e-3f * ((float)rand() / RAND_MAX - 0.5f);
      CK(hipMemcpy(df, pc.data(), fb, hipMemcpyHostToDevice));
      h.pf = pc; from_planar(h, pc, h.f);
    }
    float *dg1, *dg2, *dfs;
    CK(hipMalloc(&dg1, fb)); CK(hipMalloc(&dg2, fb)); CK(hipMemset(dg1, 0, fb)); CK(hipMemset(dg2, 0, fb));
    const size_t nfs = ics_conv2_fft_fspec_floats(h.g);
    CK(hipMalloc(&dfs, nfs * 4));
    IcsFftArgs f0; ics_conv_fft_fill_args(0, a0, dspec0, &f0); f0.planar = 63;
    IcsConvArgs a1 = conv_args(h, 1, dout, dg1, df, du, dut, dred);
    IcsFftArgs f1; ics_conv_fft_fill_args(1, a1, dspec1, &f1); f1.planar = 63;
    IcsConvArgs a2 = conv_args(h, 1, du, dg2, df, du, dut, dred + 16);
    CK(hipMemset(dred, 0, 1024));
    CK(ics_launch_conv_fft_args(0, f0, 0)); CK(ics_launch_conv_fft_args(1, f1, 0));
    CK(ics_launch_fft_image_spectrum(df + h.porg, h.g, dfs, 0));
    CK(ics_launch_conv2_fft(a2, dspec0, dspec1, dfs, 0));
    CK(hipDeviceSynchronize());
    std::vector<float> p1(h.pnf), p2(h.pnf);
    CK(hipMemcpy(p1.data(), dg1, fb, hipMemcpyDeviceToHost)); CK(hipMemcpy(p2.data(), dg2, fb, hipMemcpyDeviceToHost));
    double gm = 0, gd = 0;
    {
      const IcsGeom& g = h.g; const int pp = ics_ppitch(g); const size_t pl = ics_plane_floats(g);
      for (int c = 0; c < 3; ++c)
        for (int y = 0; y < g.uM; ++y)
          for (int x = 0; x < g.uN; ++x) { const size_t o = c * pl + h.porg + (size_t)y * pp + x; gm = fmax(gm, fabs(p1[o])); gd = fmax(gd, fabs((double)p1[o] - (double)p2[o])); }
    }
    uint32_t red[32];
    CK(hipMemcpy(red, dred, 128, hipMemcpyDeviceToHost));
    printf("mode 2 (A1 + A3 in one unit) against k_conv_fft<0> + k_conv_fft<1>: max |d| = %.3e of max |gradu| %.3e = %.3e   max |g| keys %.6g / %.6g   max u keys %.6g / %.6g  %s\n", gd, gm, gd / gm,
           ics_key2f(red[0]), ics_key2f(red[16]), ics_key2f(red[3]), ics_key2f(red[19]), gd / gm < 1e-3 ? "OK" : "FAIL");
    if (!(gd / gm < 1e-3)) rc = 1;
    if ((long)M * N <= 700L * 700L) {
      std::vector<float> g1(h.nf, 0.f), g2(h.nf, 0.f);
      from_planar(h, p1, g1); from_planar(h, p2, g2);
      std::vector<double> gref;
      reference_conv2(h, gref);
      double w1, w2;
      const double r1 = check_conv2(h, g1, gref, &w1), r2 = check_conv2(h, g2, gref, &w2);
      printf("  against float64 (relative to max |gradu|): the two kernels %.3e, mode 2 %.3e  %s\n", r1, r2, r2 < 4 * r1 + 1e-6 ? "OK" : "FAIL");
      if (!(r2 < 4 * r1 + 1e-6)) rc = 1;
    }
    for (int i = 0; i < 3; ++i) CK(ics_launch_conv2_fft(a2, dspec0, dspec1, dfs, 0));
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) CK(ics_launch_conv2_fft(a2, dspec0, dspec1, dfs, 0));
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float m2; CK(hipEventElapsedTime(&m2, e0, e1));
    IcsFftArgs f2; ics_conv_fft_fill_args(2, a2, dspec0, &f2);
    printf("  mode 2: %.4f ms per launch, %d units (%d x %d valid), %.2f us per unit and CU\n", m2 / reps, f2.nunits, f2.Vy, f2.V, 1e3 * m2 / reps / ((f2.nunits + 255) / 256));
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < 5; ++i) CK(ics_launch_fft_image_spectrum(df + h.porg, h.g, dfs, 0));
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&m2, e0, e1));
    printf("  image spectra (once per image): %.4f ms, %.1f MB\n", m2 / 5, nfs * 4 / 1e6);
  }
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < 20; ++i) CK(ics_launch_fft_spectrum(dpsf, K, dspec0, dspec1, 0));
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("  spectrum kernel: %.4f ms\n", ms / 20);
  return rc;
}

// GPU: the tap-block kernels (PSF sizes above the single-tile range) against float64 sums on a small frame, then timed on the given one
int gpu_blk(int M, int K, int N, int reps) {
  Host h = make_host(M, N, K);
  int nb, Kb;
  ics_conv_fft_blk_shape(K, &nb, &Kb);
  float *du, *de, *df, *dut, *dout, *dpsf, *dspec0, *dspec1; uint32_t* dred;
  const size_t fb = h.pnf * 4;
  for (float** p : {&du, &de, &df, &dut, &dout}) CK(hipMalloc(p, fb));
  CK(hipMalloc(&dpsf, h.psf.size() * 4)); CK(hipMalloc(&dred, 1024)); CK(hipMemset(dred, 0, 1024));
  const size_t sf = (size_t)nb * nb * ics_conv_fft_spectrum_floats();
  CK(hipMalloc(&dspec0, sf * 4)); CK(hipMalloc(&dspec1, sf * 4));
  for (size_t i = 0; i < h.nf; ++i) h.e[i] = h.e[i] + 0.01f * (h.u[i] - 0.5f) * (h.f[i] != 0.f);
  to_planar(h, h.e, h.pe);
  CK(hipMemcpy(du, h.pu.data(), fb, hipMemcpyHostToDevice)); CK(hipMemcpy(de, h.pe.data(), fb, hipMemcpyHostToDevice));
  CK(hipMemcpy(df, h.pf.data(), fb, hipMemcpyHostToDevice)); CK(hipMemcpy(dut, h.put.data(), fb, hipMemcpyHostToDevice));
  CK(hipMemcpy(dpsf, h.psf.data(), h.psf.size() * 4, hipMemcpyHostToDevice));
  CK(ics_launch_fft_spectrum(dpsf, K, dspec0, dspec1, 0, nb, Kb));
  CK(hipDeviceSynchronize());
  int rc = 0;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const bool small = (long)M * N <= 300L * 300L;
  for (int mode = 0; mode < 2; ++mode) {
    CK(hipMemset(dout, 0, fb));
    IcsConvArgs a = conv_args(h, mode, mode == 0 ? du : de, dout, df, du, dut, dred);
    CK(ics_launch_conv_fft_blk(mode, a, mode == 0 ? dspec0 : dspec1, nb, Kb, 0));
    CK(hipDeviceSynchronize());
    if (small) {
      std::vector<float> out(h.nf, 0.f), po(h.pnf);
      CK(hipMemcpy(po.data(), dout, fb, hipMemcpyDeviceToHost)); from_planar(h, po, out);
      double wa;
      const double rel = check(h, mode, out, 1, &wa);
      printf("GPU %d x %d, K = %d as %d x %d tap blocks of %d, mode %d: max |d| = %.3e, relative to max |conv| = %.3e  %s\n", M, N, K, nb, nb, Kb, mode, wa, rel, rel < 5e-6 ? "OK" : "FAIL");
      if (!(rel < 5e-6)) rc = 1;
    }
    for (int i = 0; i < 2; ++i) CK(ics_launch_conv_fft_blk(mode, a, mode == 0 ? dspec0 : dspec1, nb, Kb, 0));
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) CK(ics_launch_conv_fft_blk(mode, a, mode == 0 ? dspec0 : dspec1, nb, Kb, 0));
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("  mode %d: %.4f ms per launch\n", mode, ms / reps);
  }
  {
    float *dpart, *dgk;
    CK(hipMalloc(&dpart, (size_t)768 * Kb * Kb * 4)); CK(hipMalloc(&dgk, (size_t)3 * K * K * 4)); CK(hipMemset(dgk, 0, (size_t)3 * K * K * 4));
    CK(ics_launch_gradk_fft_blk(du + h.porg, de + h.porg, h.g, nb, Kb, dpart, dgk, 0));
    CK(hipDeviceSynchronize());
    if (small) {
      std::vector<float> gk((size_t)3 * K * K);
      CK(hipMemcpy(gk.data(), dgk, gk.size() * 4, hipMemcpyDeviceToHost));
      const double rel = check_gradk(h, h.e, gk, 5);
      printf("GPU %d x %d, K = %d, PSF gradient as %d x %d lag blocks: relative to max |gradk| = %.3e  %s\n", M, N, K, nb, nb, rel, rel < 1e-5 ? "OK" : "FAIL");
      if (!(rel < 1e-5)) rc = 1;
    }
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) CK(ics_launch_gradk_fft_blk(du + h.porg, de + h.porg, h.g, nb, Kb, dpart, dgk, 0));
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("  PSF gradient, %d launches: %.4f ms\n", nb * nb, ms / reps);
  }
  return rc;
}

}  // namespace

int main(int argc, char** argv) {
  if (argc > 1 && !strcmp(argv[1], "emulate")) {
    const int M = argc > 2 ? atoi(argv[2]) : 150, K = argc > 3 ? atoi(argv[3]) : 31, N = argc > 4 ? atoi(argv[4]) : 170;
    if (ics_conv_fft_blk_supported(K)) return emulate_blk(M, K, N);
    return emulate(M, K, N) | emulate_gradk(M, K, N) | emulate_fused(M, K, N) | (128 - 2 * K + 2 >= 16 ? emulate_conv2(M, K, N) : 0);
  }
  const int M = argc > 1 ? atoi(argv[1]) : 6144, K = argc > 2 ? atoi(argv[2]) : 31, N = argc > 3 ? atoi(argv[3]) : M, reps = argc > 4 ? atoi(argv[4]) : 20;
  if (ics_conv_fft_blk_supported(K)) return gpu_blk(M, K, N, reps);
  return gpu(M, K, N, reps);
}
