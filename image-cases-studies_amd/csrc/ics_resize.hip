// ics_resize.hip -- bicubic resize between pyramid levels (reference deconvolve.py:245-249:
// skimage.transform.resize(img, shape, order=3, mode="edge")), float64 like the SciPy code skimage runs on.
//   1. when shrinking, a separable Gaussian (sigma = (scale - 1) / 2, radius int(4 sigma + 0.5)), edge-replicated
//   2. edge padding by 12 samples and the cubic B-spline prefilter (pole sqrt(3) - 2, gain 6, mirror
//      initialisation) along y, then -- after a transpose, so that both passes run with one thread per line and
//      coalesced accesses across lines -- along x
//   3. evaluation of the spline at the pixel-centre grid  y = (i + 0.5) H / OH - 0.5
// The algorithm is written out and pinned against scipy.ndimage in oracle/resize_oracle.py (skimage itself is absent
// from the image: parity with the reference's resize is unpinned).  Not on the hot path: one call per pyramid level.
#include "ics_kernels.h"

namespace {

constexpr int NPAD = 12;

// out[y][x][c] = sum_k w[k] * in[clamp(y + k - r)][x][c]   (AXIS 0)   or along x (AXIS 1); the sum of a value runs over k = -r .. r in
// that order in every form below.
// The first version was a flat loop: one value per thread, 2 r + 1 loads of the input and of the weights (global, uniform) per value --
// bound by the L2 (13 taps x 0.4 GB per call at 4096^2) and by the weight loads' latency: 0.41 / 0.35 ms per call.  Now the weights sit
// in LDS; along y a thread keeps RS_GY consecutive rows of one column in flight and loads every input once for all of them
// (RS_GY + 2 r loads for RS_GY values); along x a workgroup stages its run of the row (+ r pixels either side) in LDS.
constexpr int RS_GY = 8, RS_GTAPS = 64;
template <typename InT>
__global__ __launch_bounds__(256) void k_rs_gauss_y(const InT* __restrict__ in, double* __restrict__ out, int H, int W, int C,
                                                    const double* __restrict__ w, int r) {
  __shared__ double sw[RS_GTAPS];
  for (int k = threadIdx.x; k < 2 * r + 1; k += 256) sw[k] = w[k];
  __syncthreads();
  const int rowl = W * C;
  const int y0 = blockIdx.y * RS_GY;
  for (int xc = blockIdx.x * 256 + threadIdx.x; xc < rowl; xc += gridDim.x * 256) {
    double acc[RS_GY];
#pragma unroll
    for (int j = 0; j < RS_GY; ++j) acc[j] = 0.0;
    for (int t = y0 - r; t < y0 + RS_GY + r; ++t) {          // unclamped row index: an edge row counts once per tap that lands on it
      const int yy = t < 0 ? 0 : (t > H - 1 ? H - 1 : t);
      const double v = (double)in[(long)yy * rowl + xc];
#pragma unroll
      for (int j = 0; j < RS_GY; ++j) {
        const int k = t - (y0 + j) + r;                        // tap of output row y0 + j that reads row t
        if (k >= 0 && k <= 2 * r) acc[j] += sw[k] * v;
      }
    }
#pragma unroll
    for (int j = 0; j < RS_GY; ++j)
      if (y0 + j < H) out[(long)(y0 + j) * rowl + xc] = acc[j];
  }
}
template <typename InT>
__global__ __launch_bounds__(256) void k_rs_gauss_x(const InT* __restrict__ in, double* __restrict__ out, int H, int W, int C,
                                                    const double* __restrict__ w, int r) {
  extern __shared__ double sm[];                 // [RS_GTAPS] weights, then the staged run: (256 + 2 r C) values
  double* sw = sm;
  double* run = sm + RS_GTAPS;
  for (int k = threadIdx.x; k < 2 * r + 1; k += 256) sw[k] = w[k];
  const int rowl = W * C;
  const int y = blockIdx.y;
  const InT* row = in + (long)y * rowl;
  for (int x0c = blockIdx.x * 256; x0c < rowl; x0c += gridDim.x * 256) {
    __syncthreads();
    // values x0c - r C .. x0c + 255 + r C of the row, clamped per PIXEL (channel kept)
    for (int e = threadIdx.x; e < 256 + 2 * r * C; e += 256) {
      const int f = x0c - r * C + e;
      const int c = ((f % C) + C) % C;
      int x = (f - c) / C;
      x = x < 0 ? 0 : (x > W - 1 ? W - 1 : x);
      run[e] = (double)row[x * C + c];
    }
    __syncthreads();
    const int xc = x0c + threadIdx.x;
    if (xc < rowl) {
      double s = 0.0;
      for (int k = 0; k <= 2 * r; ++k) s += sw[k] * run[threadIdx.x + k * C];
      out[(long)y * rowl + xc] = s;
    }
  }
}
// (radii beyond the LDS table: the flat form)
template <int AXIS, typename InT>
__global__ __launch_bounds__(256) void k_rs_gauss(const InT* __restrict__ in, double* __restrict__ out, int H, int W, int C,
                                                  const double* __restrict__ w, int r) {
  const int rowl = W * C;
  const int y = blockIdx.y;
  for (int xc = blockIdx.x * 256 + threadIdx.x; xc < rowl; xc += gridDim.x * 256) {
    const int x = xc / C, c = xc - x * C;
    double s = 0.0;
    for (int k = -r; k <= r; ++k) {
      int yy = y, xx = x;
      if (AXIS == 0) { yy = y + k; yy = yy < 0 ? 0 : (yy > H - 1 ? H - 1 : yy); }
      else { xx = x + k; xx = xx < 0 ? 0 : (xx > W - 1 ? W - 1 : xx); }
      s += w[k + r] * (double)in[(long)yy * rowl + xx * C + c];
    }
    out[(long)y * rowl + xc] = s;
  }
}

// edge padding by NPAD on both spatial axes (one padded row per blockIdx.y)
template <typename InT>
__global__ __launch_bounds__(256) void k_rs_pad(const InT* __restrict__ in, double* __restrict__ out, int H, int W, int C) {
  const int Wp = W + 2 * NPAD, rowl = Wp * C;
  int y = (int)blockIdx.y - NPAD;
  y = y < 0 ? 0 : (y > H - 1 ? H - 1 : y);
  for (int xc = blockIdx.x * 256 + threadIdx.x; xc < rowl; xc += gridDim.x * 256) {
    const int xq = xc / C, c = xc - xq * C;
    int x = xq - NPAD;
    x = x < 0 ? 0 : (x > W - 1 ? W - 1 : x);
    out[(long)blockIdx.y * rowl + xc] = (double)in[((long)y * W + x) * C + c];
  }
}

// Cubic B-spline prefilter along the slow axis of a[n][L] (L independent lines): scipy ni_splines.c -- gain, causal mirror
// initialisation, forward recursion c+[i] = gain x[i] + z c+[i-1], anticausal initialisation, backward recursion
// c[i] = z (c[i+1] - c+[i]); pole z = sqrt(3) - 2.
// Round 4: the first version ran one thread per line through all n samples, three dependent passes of strided accesses: 6144 threads
// on 24 CUs, 0.8 ms per call on average and 36 ms of a 190-ms device-resident 2048^2 deblur_module run (profiles/r04_driver_trace_before.txt).
// The recursions forget: |z|^40 = 1.3e-23, far below the 1.1e-16 of a double.  So a line is cut into segments of SEG samples, one
// thread per (line, segment); the forward pass starts WARM samples before its segment from state 0 (the first segment from the
// exact mirror sum), the backward pass WARM samples behind it (the last segment from the exact anticausal value).  Two out-of-place
// kernels (a thread's warm-up reads what another thread owns): forward in -> out, backward out -> in.  Against the sequential form the
// results agree to the last bit wherever the discarded history is below half an ulp, i.e. everywhere but on rounding ties
// (tests/test_resize.py pins the whole resize within 1e-12 of scipy.ndimage).  The mirror sum of the first sample runs over 120
// terms (|z|^120 = 1e-69) instead of until z^i underflows (~540): same double for any data whose dynamic range is below 1e50.
constexpr int RS_WARM = 40, RS_HORIZON = 120;
// samples a thread owns: 32 on short lines (parallelism), more on long ones (the 40 warm-up samples are re-read per segment: 2.25 reads per
// sample at 32, 1.3 at 128)
static inline int rs_seg(int n) { return n >= 4096 ? 128 : (n >= 1024 ? 64 : 32); }
constexpr double RS_Z = -0.26794919243112270647;   // sqrt(3) - 2

__global__ __launch_bounds__(256) void k_rs_prefilter_fwd(const double* __restrict__ in, double* __restrict__ out, int n, long L, int RS_SEG) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  const int nseg = (n + RS_SEG - 1) / RS_SEG;
  if (t >= L * nseg) return;
  const long l = t % L;                       // neighbouring threads = neighbouring lines: coalesced
  const int sgm = (int)(t / L);
  const int a = sgm * RS_SEG, b = a + RS_SEG < n ? a + RS_SEG : n;
  const double z = RS_Z, gain = (1.0 - z) * (1.0 - 1.0 / z);
  const double* p = in + l;
  double prev;
  int i;
  if (a - RS_WARM <= 0) {                     // from the line's start: exact causal mirror initialisation
    const double zn = pow(z, (double)(n - 1));
    const double last = p[(long)(n - 1) * L] * gain;
    double c0 = p[0] * gain + zn * last, zi = z;
    const int m = n - 1 < RS_HORIZON ? n - 1 : RS_HORIZON;
    for (int k = 1; k < m; ++k) {
      c0 += zi * (p[(long)k * L] * gain + zn * (p[(long)(n - 1 - k) * L] * gain));
      zi *= z;
    }
    prev = c0 / (1.0 - zn * zn);
    if (a == 0) out[l] = prev;
    i = 1;
  } else {
    prev = 0.0;
    i = a - RS_WARM;
  }
  for (; i < a; ++i) prev = p[(long)i * L] * gain + z * prev;          // warm-up (not stored)
  for (i = i > a ? i : a; i < b; ++i) {
    prev = p[(long)i * L] * gain + z * prev;
    out[(long)i * L + l] = prev;
  }
}

__global__ __launch_bounds__(256) void k_rs_prefilter_bwd(const double* __restrict__ cp, double* __restrict__ out, int n, long L, int RS_SEG) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  const int nseg = (n + RS_SEG - 1) / RS_SEG;
  if (t >= L * nseg) return;
  const long l = t % L;
  const int sgm = (int)(t / L);
  const int a = sgm * RS_SEG, b = a + RS_SEG < n ? a + RS_SEG : n;
  const double z = RS_Z;
  const double* p = cp + l;
  double nxt;
  int i;
  if (b + RS_WARM >= n) {                     // to the line's end: exact anticausal initialisation
    nxt = (z * p[(long)(n - 2) * L] + p[(long)(n - 1) * L]) * z / (z * z - 1.0);
    if (b == n) out[(long)(n - 1) * L + l] = nxt;
    i = n - 2;
  } else {
    nxt = 0.0;
    i = b + RS_WARM - 1;
  }
  for (; i >= b; --i) nxt = z * (nxt - p[(long)i * L]);                 // warm-up
  for (i = i < b - 1 ? i : b - 1; i >= a; --i) {
    nxt = z * (nxt - p[(long)i * L]);
    out[(long)i * L + l] = nxt;
  }
}

// out[x][y][c] = in[y][x][c]  (32 x 32 pixel tiles through LDS).  C = 3: a tile row is a run of 96 consecutive doubles on both sides -- the
// first version moved one channel per workgroup with a 24-byte stride between lanes, i.e. read and wrote every cache line three times
// (0.42 ms for a 4096^2 frame).  Other channel counts keep that form.
__global__ __launch_bounds__(256) void k_rs_transpose3(const double* __restrict__ in, double* __restrict__ out, int H, int W) {
  __shared__ double tile[32][97];
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
  const int nx = W - bx < 32 ? W - bx : 32, ny = H - by < 32 ? H - by : 32;     // pixels of this tile
  for (int e = threadIdx.x; e < 32 * 96; e += 256) {
    const int row = e / 96, col = e - row * 96;
    if (row < ny && col < 3 * nx) tile[row][col] = in[((long)(by + row) * W + bx) * 3 + col];
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 32 * 96; e += 256) {
    const int rx = e / 96, col = e - rx * 96;              // output row = x, within it the tile's y pixels
    const int ry = col / 3, c = col - 3 * ry;
    if (rx < nx && ry < ny) out[((long)(bx + rx) * H + by) * 3 + col] = tile[ry][3 * rx + c];
  }
}
__global__ __launch_bounds__(256) void k_rs_transpose(const double* __restrict__ in, double* __restrict__ out, int H, int W, int C) {
  __shared__ double tile[32][33];
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32, c = blockIdx.z;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int k = ty; k < 32; k += 8) {
    const int y = by + k, x = bx + tx;
    tile[k][tx] = (y < H && x < W) ? in[((long)y * W + x) * C + c] : 0.0;
  }
  __syncthreads();
  for (int k = ty; k < 32; k += 8) {
    const int x = bx + k, y = by + tx;
    if (x < W && y < H) out[((long)x * H + y) * C + c] = tile[tx][k];
  }
}

__device__ __forceinline__ void bspline3(double t, double (&w)[4]) {
  const double t2 = t * t, t3 = t2 * t, u = 1.0 - t;
  w[0] = u * u * u / 6.0;
  w[1] = (3.0 * t3 - 6.0 * t2 + 4.0) / 6.0;
  w[2] = (-3.0 * t3 + 3.0 * t2 + 3.0 * t + 1.0) / 6.0;
  w[3] = t3 / 6.0;
}

// coefT[xp][yp][c] (transposed, padded) -> out[i][j][c].  A workgroup evaluates a tile of 32 x 32 output pixels: its lanes run along the
// output ROWS i (and channels) first, which is the fast axis of the transposed coefficients, and the results leave through LDS as runs of
// 96 consecutive doubles.  (The first version ran the lanes along j: 16 gathered reads per value at a stride of a whole coefficient line,
// and two 64-bit divisions -- 0.37 ms for a 4096^2 frame.)  The sum of a value is formed in the same order as before.
template <typename OutT>
__global__ __launch_bounds__(256) void k_rs_eval(const double* __restrict__ coefT, int Hp, int Wp, int C, double fy, double fx,
                                                 OutT* __restrict__ out, int OH, int OW) {
  __shared__ double res[32][97];
  const int j0 = blockIdx.x * 32, i0 = blockIdx.y * 32;
  const bool tiled = C == 3;
  const int per = tiled ? 32 * 96 : 0;
  for (int e = threadIdx.x; e < per; e += 256) {
    const int jl = e / 96, rr = e - jl * 96;               // lanes: (i, c) fastest, then j
    const int il = rr / 3, c = rr - 3 * il;
    const int oi = i0 + il, oj = j0 + jl;
    if (oi >= OH || oj >= OW) continue;
    const double ys = ((double)oi + 0.5) * fy - 0.5 + (double)NPAD, xs = ((double)oj + 0.5) * fx - 0.5 + (double)NPAD;
    const double yf = floor(ys), xf = floor(xs);
    double wy[4], wx[4];
    bspline3(ys - yf, wy);
    bspline3(xs - xf, wx);
    const int y0 = (int)yf - 1, x0 = (int)xf - 1;
    double s = 0.0;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      int yy = y0 + a; yy = yy < 0 ? 0 : (yy > Hp - 1 ? Hp - 1 : yy);
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        int xx = x0 + b; xx = xx < 0 ? 0 : (xx > Wp - 1 ? Wp - 1 : xx);
        s += (wy[a] * wx[b]) * coefT[((long)xx * Hp + yy) * 3 + c];
      }
    }
    res[il][3 * jl + c] = s;
  }
  if (tiled) {
    __syncthreads();
    const int nj = OW - j0 < 32 ? OW - j0 : 32;
    for (int e = threadIdx.x; e < 32 * 96; e += 256) {
      const int il = e / 96, col = e - il * 96;
      if (i0 + il < OH && col < 3 * nj) out[((long)(i0 + il) * OW + j0) * 3 + col] = (OutT)res[il][col];
    }
    return;
  }
  // other channel counts: one value per thread over the tile, as the first version
  for (int e = threadIdx.x; e < 32 * 32 * C; e += 256) {
    const int c = e % C, pl = e / C, jl = pl & 31, il = pl >> 5;
    const int oi = i0 + il, oj = j0 + jl;
    if (oi >= OH || oj >= OW) continue;
    const double ys = ((double)oi + 0.5) * fy - 0.5 + (double)NPAD, xs = ((double)oj + 0.5) * fx - 0.5 + (double)NPAD;
    const double yf = floor(ys), xf = floor(xs);
    double wy[4], wx[4];
    bspline3(ys - yf, wy);
    bspline3(xs - xf, wx);
    const int y0 = (int)yf - 1, x0 = (int)xf - 1;
    double s = 0.0;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      int yy = y0 + a; yy = yy < 0 ? 0 : (yy > Hp - 1 ? Hp - 1 : yy);
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        int xx = x0 + b; xx = xx < 0 ? 0 : (xx > Wp - 1 ? Wp - 1 : xx);
        s += (wy[a] * wx[b]) * coefT[((long)xx * Hp + yy) * C + c];
      }
    }
    out[((long)oi * OW + oj) * C + c] = (OutT)s;
  }
}

inline unsigned grid_for(long n) { long b = (n + 255) / 256; return (unsigned)(b > 65536 ? 65536 : (b < 1 ? 1 : b)); }

}  // namespace

size_t ics_resize_scratch_doubles(int H, int W, int C) {   // two padded buffers (the first doubles as the Gaussian temporary)
  return 2 * (size_t)(H + 2 * NPAD) * (W + 2 * NPAD) * C;
}

// src: H x W x C on the device (overwritten by the Gaussian when shrinking); wy / wx: Gaussian weights on the device (radius ry / rx,
// NULL = no smoothing along that axis); scratch: ics_resize_scratch_doubles(); out: OH x OW x C
template <typename InT, typename OutT>
static hipError_t launch_resize_t(const InT* src, int H, int W, int C, const double* wy, int ry, const double* wx, int rx, double* scratch,
                                  OutT* out, int OH, int OW, hipStream_t s) {
  const int Hp = H + 2 * NPAD, Wp = W + 2 * NPAD;
  double* A = scratch;
  double* B = scratch + (size_t)Hp * Wp * C;
  // the first pass reads the source in its own type (float32 images: no conversion pass in front), everything behind it is float64
  const unsigned gx = (unsigned)((W * C + 255) / 256 > 64 ? 64 : (W * C + 255) / 256);
  const dim3 g_rows(gx, (unsigned)H), g_y(gx, (unsigned)((H + RS_GY - 1) / RS_GY));
  const size_t lds_x = (size_t)(RS_GTAPS + 256 + 2 * rx * C) * sizeof(double);
  double* cur = nullptr;                         // nullptr: still the source
  if (wy) {
    if (2 * ry + 1 <= RS_GTAPS) hipLaunchKernelGGL(k_rs_gauss_y<InT>, g_y, dim3(256), 0, s, src, A, H, W, C, wy, ry);
    else hipLaunchKernelGGL((k_rs_gauss<0, InT>), g_rows, dim3(256), 0, s, src, A, H, W, C, wy, ry);
    cur = A;
  }
  if (wx) {
    double* dst = (cur == A) ? B : A;
    if (cur) {
      if (2 * rx + 1 <= RS_GTAPS) hipLaunchKernelGGL(k_rs_gauss_x<double>, g_rows, dim3(256), lds_x, s, (const double*)cur, dst, H, W, C, wx, rx);
      else hipLaunchKernelGGL((k_rs_gauss<1, double>), g_rows, dim3(256), 0, s, (const double*)cur, dst, H, W, C, wx, rx);
    } else {
      if (2 * rx + 1 <= RS_GTAPS) hipLaunchKernelGGL(k_rs_gauss_x<InT>, g_rows, dim3(256), lds_x, s, src, dst, H, W, C, wx, rx);
      else hipLaunchKernelGGL((k_rs_gauss<1, InT>), g_rows, dim3(256), 0, s, src, dst, H, W, C, wx, rx);
    }
    cur = dst;
  }
  // pad into the buffer that does not hold `cur`; from there the two buffers alternate (the prefilter passes are out of place)
  double* P = (cur == A) ? B : A;
  double* T = (P == A) ? B : A;
  auto pf_grid = [](int n, long L) { const int sg = rs_seg(n); return dim3((unsigned)((L * ((n + sg - 1) / sg) + 255) / 256)); };
  const dim3 g_pad((unsigned)((Wp * C + 255) / 256 > 64 ? 64 : (Wp * C + 255) / 256), (unsigned)Hp);
  if (cur) hipLaunchKernelGGL(k_rs_pad<double>, g_pad, dim3(256), 0, s, (const double*)cur, P, H, W, C);
  else hipLaunchKernelGGL(k_rs_pad<InT>, g_pad, dim3(256), 0, s, src, P, H, W, C);
  hipLaunchKernelGGL(k_rs_prefilter_fwd, pf_grid(Hp, (long)Wp * C), dim3(256), 0, s, P, T, Hp, (long)Wp * C, rs_seg(Hp));
  hipLaunchKernelGGL(k_rs_prefilter_bwd, pf_grid(Hp, (long)Wp * C), dim3(256), 0, s, T, P, Hp, (long)Wp * C, rs_seg(Hp));
  if (C == 3) hipLaunchKernelGGL(k_rs_transpose3, dim3((Wp + 31) / 32, (Hp + 31) / 32), dim3(256), 0, s, P, T, Hp, Wp);
  else hipLaunchKernelGGL(k_rs_transpose, dim3((Wp + 31) / 32, (Hp + 31) / 32, C), dim3(256), 0, s, P, T, Hp, Wp, C);
  hipLaunchKernelGGL(k_rs_prefilter_fwd, pf_grid(Wp, (long)Hp * C), dim3(256), 0, s, T, P, Wp, (long)Hp * C, rs_seg(Wp));
  hipLaunchKernelGGL(k_rs_prefilter_bwd, pf_grid(Wp, (long)Hp * C), dim3(256), 0, s, P, T, Wp, (long)Hp * C, rs_seg(Wp));
  hipLaunchKernelGGL(k_rs_eval<OutT>, dim3((OW + 31) / 32, (OH + 31) / 32), dim3(256), 0, s, T, Hp, Wp, C, (double)H / (double)OH, (double)W / (double)OW, out, OH, OW);
  return hipGetLastError();
}

hipError_t ics_launch_resize(double* src, int H, int W, int C, const double* wy, int ry, const double* wx, int rx, double* scratch,
                             double* out, int OH, int OW, hipStream_t s) {
  return launch_resize_t<double, double>(src, H, W, C, wy, ry, wx, rx, scratch, out, OH, OW, s);
}
// float32 images (ics_img_resize): read and written in place of the float32 <-> float64 conversion passes around the float64 pipeline
hipError_t ics_launch_resize_f32(const float* src, int H, int W, int C, const double* wy, int ry, const double* wx, int rx, double* scratch,
                                 float* out, int OH, int OW, hipStream_t s) {
  return launch_resize_t<float, float>(src, H, W, C, wy, ry, wx, rx, scratch, out, OH, OW, s);
}
