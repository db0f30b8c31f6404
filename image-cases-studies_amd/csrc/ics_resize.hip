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

// out[y][x][c] = sum_k w[k] * in[clamp(y + k - r)][x][c]   (AXIS 0)   or along x (AXIS 1)
template <int AXIS>
__global__ __launch_bounds__(256) void k_rs_gauss(const double* __restrict__ in, double* __restrict__ out, int H, int W, int C,
                                                  const double* __restrict__ w, int r) {
  const long n = (long)H * W * C;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C);
    const long p = i / C;
    const int x = (int)(p % W), y = (int)(p / W);
    double s = 0.0;
    for (int k = -r; k <= r; ++k) {
      int yy = y, xx = x;
      if (AXIS == 0) { yy = y + k; yy = yy < 0 ? 0 : (yy > H - 1 ? H - 1 : yy); }
      else { xx = x + k; xx = xx < 0 ? 0 : (xx > W - 1 ? W - 1 : xx); }
      s += w[k + r] * in[((long)yy * W + xx) * C + c];
    }
    out[i] = s;
  }
}

// edge padding by NPAD on both spatial axes
__global__ __launch_bounds__(256) void k_rs_pad(const double* __restrict__ in, double* __restrict__ out, int H, int W, int C) {
  const int Hp = H + 2 * NPAD, Wp = W + 2 * NPAD;
  const long n = (long)Hp * Wp * C;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C);
    const long p = i / C;
    int x = (int)(p % Wp) - NPAD, y = (int)(p / Wp) - NPAD;
    x = x < 0 ? 0 : (x > W - 1 ? W - 1 : x);
    y = y < 0 ? 0 : (y > H - 1 ? H - 1 : y);
    out[i] = in[((long)y * W + x) * C + c];
  }
}

// cubic B-spline prefilter along the slow axis of a[n][L] (L independent lines, one thread each; neighbouring threads
// touch neighbouring addresses).  scipy ni_splines.c: gain, causal mirror initialisation (exact sum), forward recursion,
// anticausal initialisation, backward recursion.
__global__ __launch_bounds__(256) void k_rs_prefilter(double* __restrict__ a, int n, long L) {
  const long l = (long)blockIdx.x * 256 + threadIdx.x;
  if (l >= L) return;
  const double z = -0.26794919243112270647;   // sqrt(3) - 2
  const double gain = (1.0 - z) * (1.0 - 1.0 / z);
  double* p = a + l;
  const double zn = pow(z, (double)(n - 1));
  const double last = p[(long)(n - 1) * L] * gain;
  double c0 = p[0] * gain + zn * last;
  double zi = z;
  for (int i = 1; i < n - 1; ++i) {
    c0 += zi * (p[(long)i * L] * gain + zn * (p[(long)(n - 1 - i) * L] * gain));
    zi *= z;
    if (zi == 0.0) break;   // underflow: every further term is exactly zero
  }
  double prev = c0 / (1.0 - zn * zn);
  p[0] = prev;
  for (int i = 1; i < n; ++i) {
    prev = p[(long)i * L] * gain + z * prev;
    p[(long)i * L] = prev;
  }
  double nxt = (z * p[(long)(n - 2) * L] + p[(long)(n - 1) * L]) * z / (z * z - 1.0);
  p[(long)(n - 1) * L] = nxt;
  for (int i = n - 2; i >= 0; --i) {
    nxt = z * (nxt - p[(long)i * L]);
    p[(long)i * L] = nxt;
  }
}

// out[x][y][c] = in[y][x][c]  (32 x 32 tiles through LDS)
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

// coefT[xp][yp][c] (transposed, padded) -> out[i][j][c]
__global__ __launch_bounds__(256) void k_rs_eval(const double* __restrict__ coefT, int Hp, int Wp, int C, double fy, double fx,
                                                 double* __restrict__ out, int OH, int OW) {
  const long n = (long)OH * OW * C;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C);
    const long p = i / C;
    const int oj = (int)(p % OW), oi = (int)(p / OW);
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
    out[i] = s;
  }
}

inline unsigned grid_for(long n) { long b = (n + 255) / 256; return (unsigned)(b > 65536 ? 65536 : (b < 1 ? 1 : b)); }

}  // namespace

size_t ics_resize_scratch_doubles(int H, int W, int C) {   // two padded buffers (the first doubles as the Gaussian temporary)
  return 2 * (size_t)(H + 2 * NPAD) * (W + 2 * NPAD) * C;
}

// src: H x W x C on the device (overwritten by the Gaussian when shrinking); wy / wx: Gaussian weights on the device (radius ry / rx,
// NULL = no smoothing along that axis); scratch: ics_resize_scratch_doubles(); out: OH x OW x C
hipError_t ics_launch_resize(double* src, int H, int W, int C, const double* wy, int ry, const double* wx, int rx, double* scratch,
                             double* out, int OH, int OW, hipStream_t s) {
  const int Hp = H + 2 * NPAD, Wp = W + 2 * NPAD;
  double* A = scratch;
  double* B = scratch + (size_t)Hp * Wp * C;
  const long n = (long)H * W * C;
  double* cur = src;
  if (wy) { hipLaunchKernelGGL(k_rs_gauss<0>, dim3(grid_for(n)), dim3(256), 0, s, cur, A, H, W, C, wy, ry); cur = A; }
  if (wx) { double* dst = (cur == A) ? B : A; hipLaunchKernelGGL(k_rs_gauss<1>, dim3(grid_for(n)), dim3(256), 0, s, cur, dst, H, W, C, wx, rx); cur = dst; }
  // pad into the buffer that does not hold `cur`
  double* P = (cur == A) ? B : A;
  double* T = (P == A) ? B : A;
  hipLaunchKernelGGL(k_rs_pad, dim3(grid_for((long)Hp * Wp * C)), dim3(256), 0, s, cur, P, H, W, C);
  hipLaunchKernelGGL(k_rs_prefilter, dim3((unsigned)(((long)Wp * C + 255) / 256)), dim3(256), 0, s, P, Hp, (long)Wp * C);
  hipLaunchKernelGGL(k_rs_transpose, dim3((Wp + 31) / 32, (Hp + 31) / 32, C), dim3(256), 0, s, P, T, Hp, Wp, C);
  hipLaunchKernelGGL(k_rs_prefilter, dim3((unsigned)(((long)Hp * C + 255) / 256)), dim3(256), 0, s, T, Wp, (long)Hp * C);
  hipLaunchKernelGGL(k_rs_eval, dim3(grid_for((long)OH * OW * C)), dim3(256), 0, s, T, Hp, Wp, C, (double)H / (double)OH, (double)W / (double)OW, out, OH, OW);
  return hipGetLastError();
}
