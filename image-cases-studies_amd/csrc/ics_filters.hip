// ics_filters.hip -- standalone operators next to the RL loop: the TV stencil of
// lib/deconvolution.pyx:137-239 and the lib/utils.py filters (Gaussian/Bessel blur, USM, bilateral).
// The TV stencil is 3 x 3 (one output per lane, neighbours from L1/L2); the blurs and the bilateral filter are LDS-tiled
// (k_conv2d_tile, k_bilateral_tile below).
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

// LDS-tiled: a 256-thread workgroup owns a 32 x 32 output tile; the tile plus its (KH-1) x (KW-1) halo is staged once in LDS
// with the symmetric extension resolved at load time (coalesced row segments), every lane then produces 4 outputs (rows
// ty, ty+8, ty+16, ty+24 of column tx) from LDS: KH*KW LDS reads per output instead of KH*KW global reads, and the weights
// are wave-uniform (scalar loads).  float64 like SciPy.  Rank-1 kernels -- every lib/utils.py window is an outer product --
// run as two passes of this kernel (1 x KW, then KH x 1): 2K instead of K^2 taps; the USM epilogue (lib/utils.py:275,
// src + (src - blur) * amount) reads the ORIGINAL channel `src0`.
#define CT 32
__global__ __launch_bounds__(256) void k_conv2d_tile(const double* __restrict__ src, int H, int W, const double* __restrict__ kern,
                                                    int KH, int KW, double* __restrict__ out, const double* __restrict__ src0, int usm, double amount) {
  extern __shared__ __attribute__((aligned(16))) double tile[];
  const int cy = (KH - 1) / 2, cx = (KW - 1) / 2;
  const int LW = CT + KW - 1, LH = CT + KH - 1;
  const int x0 = blockIdx.x * CT, y0 = blockIdx.y * CT;
  // "same": out[i][j] = sum_{p,q} kern[p][q] * ext[i + cy - p][j + cx - q]  ->  staged rows y0 + cy - (KH-1) .. y0 + cy + CT - 1
  const int ys = y0 + cy - (KH - 1), xs = x0 + cx - (KW - 1);
  for (int t = threadIdx.x; t < LH * LW; t += 256) {
    const int r = t / LW, c = t - r * LW;
    tile[t] = src[(long)symm(ys + r, H) * W + symm(xs + c, W)];
  }
  __syncthreads();
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  for (int p = 0; p < KH; ++p) {
    for (int q = 0; q < KW; ++q) {
      const double w = kern[p * KW + q];                       // uniform
      // tile row of output row ty + 8m for tap p: (ty + 8m) + (KH-1) - p ; column tx + (KW-1) - q
      const double* tp = tile + (ty + KH - 1 - p) * LW + (tx + KW - 1 - q);
#pragma unroll
      for (int m = 0; m < 4; ++m) acc[m] += w * tp[8 * m * LW];
    }
  }
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int y = y0 + ty + 8 * m, x = x0 + tx;
    if (y < H && x < W) {
      const long o = (long)y * W + x;
      out[o] = usm ? src0[o] + (src0[o] - acc[m]) * amount : acc[m];
    }
  }
}

// ---- bilateral filter (lib/utils.py:173-234), gaussian(x, s) = exp(-x^2 / (2 s^2)) --------------
// Same tiling: the 32 x 32 tile + radius halo (symmetric padding, lib/utils.py:204) in LDS; the spatial weights
// exp(-(i^2 + j^2) / 2 std_s^2) of the (2r+1)^2 offsets are precomputed once per call (`ws`, wave-uniform reads), so each tap
// costs one float64 exp (the range term) instead of two.  Accumulation order = the reference's offset order (j slow, i fast).
__global__ __launch_bounds__(256) void k_bilateral_tile(const double* __restrict__ src, int H, int W, int radius, double std_i,
                                                       const double* __restrict__ ws, double* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) double tile[];
  const int D = 2 * radius + 1, LW = CT + 2 * radius, LH = CT + 2 * radius;
  const int x0 = blockIdx.x * CT, y0 = blockIdx.y * CT;
  for (int t = threadIdx.x; t < LH * LW; t += 256) {
    const int r = t / LW, c = t - r * LW;
    tile[t] = src[(long)symm(y0 - radius + r, H) * W + symm(x0 - radius + c, W)];
  }
  __syncthreads();
  const double ki = -1.0 / (2.0 * std_i * std_i);
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  double cen[4], acc[4] = {0.0, 0.0, 0.0, 0.0}, wsum[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int m = 0; m < 4; ++m) cen[m] = tile[(ty + 8 * m + radius) * LW + tx + radius];
  for (int j = -radius; j <= radius; ++j)         // lib/utils.py:209-213: j (x offset) is the slow index of `combi`
    for (int i = -radius; i <= radius; ++i) {
      const double wsp = ws[(j + radius) * D + (i + radius)];
      const double* tp = tile + (ty + radius + i) * LW + (tx + radius + j);
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const double nb = tp[8 * m * LW];
        const double w = exp((nb - cen[m]) * (nb - cen[m]) * ki) * wsp;
        acc[m] += nb * w; wsum[m] += w;
      }
    }
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int y = y0 + ty + 8 * m, x = x0 + tx;
    if (y < H && x < W) out[(long)y * W + x] = acc[m] / wsum[m];
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

static hipError_t set_lds(const void* kern, size_t bytes) {
  if (bytes <= 64 * 1024) return hipSuccess;
  return hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

// one pass of the tiled convolution: out = conv2d_symm(src, kern) [USM epilogue against src0]
hipError_t ics_launch_conv2d_symm(const double* src, int H, int W, const double* kern, int KH, int KW, double* out,
                                  const double* src0, int usm, double amount, hipStream_t s) {
  const size_t lds = (size_t)(CT + KH - 1) * (CT + KW - 1) * sizeof(double);
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  hipError_t e = set_lds(reinterpret_cast<const void*>(k_conv2d_tile), lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_conv2d_tile, dim3((W + CT - 1) / CT, (H + CT - 1) / CT), dim3(256), lds, s, src, H, W, kern, KH, KW, out, src0, usm, amount);
  return hipGetLastError();
}

hipError_t ics_launch_bilateral(const double* src, int H, int W, int radius, double std_i, const double* ws, double* out, hipStream_t s) {
  const size_t lds = (size_t)(CT + 2 * radius) * (CT + 2 * radius) * sizeof(double);
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  hipError_t e = set_lds(reinterpret_cast<const void*>(k_bilateral_tile), lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_bilateral_tile, dim3((W + CT - 1) / CT, (H + CT - 1) / CT), dim3(256), lds, s, src, H, W, radius, std_i, ws, out);
  return hipGetLastError();
}
