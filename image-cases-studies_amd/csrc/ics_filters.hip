// ics_filters.hip -- standalone operators next to the RL loop: the TV stencil of
// lib/deconvolution.pyx:137-239 and the lib/utils.py filters (Gaussian/Bessel blur, USM, bilateral).
// All of them are small 2-D stencils on contiguous host-shaped arrays: one output element per lane,
// neighbours served by L1/L2 (3x3 .. (2r+1)^2 taps, rows are contiguous so a wave reads 64
// consecutive elements per tap).
#include "ics_kernels.h"
#include "ics_tv.h"

namespace {

// ---- TV (lib/deconvolution.pyx:137-239) ---------------------------------------------------------
// order 2: udx = -2u + u[i-1] + u[i+1], udy likewise in j, the two diagonals divided by sqrt(2);
//          div = (-udx - udy - udxdy - udydx)/adjust;  out = (|.|(udx,udy) + |.|(udxdy,udydx))/adjust
// order 1: backward and forward first differences, four norms.
// norm 1: |x|+|y|+eps, adjust = 4(1+1/sqrt2);  norm 2: sqrt(x^2+y^2+eps^2), adjust = 2(1+sqrt2).
// Borders are left untouched (pyx:239).  Separately rounded float32 operations, reference order.
__global__ __launch_bounds__(256) void k_tv(const float* __restrict__ u, int M, int N, float eps, int order, int norm,
                                           float* __restrict__ out, float* __restrict__ dv) {
  const long total = (long)(M - 2) * (N - 2) * 3;
  const long rs = (long)N * 3;
  for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
    const int k = (int)(t % 3);
    const int j = 1 + (int)((t / 3) % (N - 2));
    const int i = 1 + (int)(t / (3L * (N - 2)));
    const long o = (long)i * rs + 3L * j + k;
    const IcsTvOut r = ics_tv_point(u[o], u[o - rs], u[o + rs], u[o - 3], u[o + 3], u[o - rs - 3], u[o + rs + 3],
                                    u[o - rs + 3], u[o + rs - 3], eps, order, norm);
    dv[o] = r.div;
    out[o] = r.out;
  }
}

// ---- scipy.signal.convolve2d(src, kern, mode="same", boundary="symm") (lib/utils.py:243-262) ----
__device__ __forceinline__ int symm(int i, int n) {  // ... x1 x0 | x0 x1 ... x(n-1) | x(n-1) x(n-2) ...
  const int p = 2 * n;
  i %= p; if (i < 0) i += p;
  return i < n ? i : p - 1 - i;
}

__global__ __launch_bounds__(256) void k_conv2d_symm(const double* __restrict__ src, int H, int W, const double* __restrict__ kern,
                                                    int KH, int KW, double* __restrict__ out, int usm, double amount) {
  const long total = (long)H * W;
  const int cy = (KH - 1) / 2, cx = (KW - 1) / 2;
  for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
    const int i = (int)(t / W), j = (int)(t - (long)i * W);
    double s = 0.0;
    for (int p = 0; p < KH; ++p) {
      const int yy = symm(i + cy - p, H);
      for (int q = 0; q < KW; ++q) s += kern[p * KW + q] * src[(long)yy * W + symm(j + cx - q, W)];
    }
    // USM (lib/utils.py:275): src + (src - blur) * amount
    out[t] = usm ? src[t] + (src[t] - s) * amount : s;
  }
}

// ---- bilateral filter (lib/utils.py:173-234), gaussian(x, s) = exp(-x^2 / (2 s^2)) --------------
__global__ __launch_bounds__(256) void k_bilateral(const double* __restrict__ src, int H, int W, int radius, double std_i, double std_s,
                                                  double* __restrict__ out) {
  const long total = (long)H * W;
  const double ki = -1.0 / (2.0 * std_i * std_i), ks = -1.0 / (2.0 * std_s * std_s);
  for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
    const int y = (int)(t / W), x = (int)(t - (long)y * W);
    const double c = src[t];
    double acc = 0.0, wsum = 0.0;
    for (int j = -radius; j <= radius; ++j)       // lib/utils.py:209-213: j is the slow index of `combi`
      for (int i = -radius; i <= radius; ++i) {
        const double nb = src[(long)symm(y + i, H) * W + symm(x + j, W)];
        const double dist2 = (double)(i * i + j * j);
        const double w = exp((nb - c) * (nb - c) * ki) * exp(dist2 * ks);
        acc += nb * w; wsum += w;
      }
    out[t] = acc / wsum;
  }
}

}  // namespace

hipError_t ics_launch_tv(const float* u, int M, int N, float eps, int order, int norm, float* out, float* div, hipStream_t s) {
  if (M < 3 || N < 3) return hipSuccess;
  const long total = (long)(M - 2) * (N - 2) * 3;
  long blocks = (total + 255) / 256; if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(k_tv, dim3((unsigned)blocks), dim3(256), 0, s, u, M, N, eps, order, norm, out, div);
  return hipGetLastError();
}

hipError_t ics_launch_conv2d_symm(const double* src, int H, int W, const double* kern, int KH, int KW, double* out,
                                  int usm, double amount, hipStream_t s) {
  const long total = (long)H * W;
  long blocks = (total + 255) / 256; if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(k_conv2d_symm, dim3((unsigned)blocks), dim3(256), 0, s, src, H, W, kern, KH, KW, out, usm, amount);
  return hipGetLastError();
}

hipError_t ics_launch_bilateral(const double* src, int H, int W, int radius, double std_i, double std_s, double* out, hipStream_t s) {
  const long total = (long)H * W;
  long blocks = (total + 255) / 256; if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(k_bilateral, dim3((unsigned)blocks), dim3(256), 0, s, src, H, W, radius, std_i, std_s, out);
  return hipGetLastError();
}
