// ics_stats.hip -- per-outer-iteration statistics and the residual-whiteness stop metric, on device.
//
//   A18 (lib/deconvolution.pyx:600-601): varu = std(u[window])^2,  Hu = ||error[window]||^2 / (n*3)
//   A19 (lib/deconvolution.pyx:627-638): t = (e - mean e)/std e;  t /= max|t|;
//        per channel  ac = convolve(t, rot180 t, "same");  M_r = mean(ac^2 * w)
// The reference evaluates the autocorrelation with scipy's FFT convolution (pyx:632).  Here the
// window (<= 1024 px) is zero-padded to a power of two P >= 2*max(H,W)-1 and autocorrelated by
// Wiener-Khinchin with a radix-2 Stockham FFT that runs entirely in LDS (one line per workgroup):
// rows of the window, columns, |Z|^2 + inverse columns, inverse rows of the lags that are read.  3*P*P complex64 = 6 MB for a 255^2 window,
// a few tens of microseconds per outer iteration, so the stop test never leaves the GPU.
#include "ics_kernels.h"

namespace {

__device__ __forceinline__ double block_sum(double v, double* sh) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  double s = 0.0;
  const int nw = blockDim.x >> 6;
  for (int i = 0; i < nw; ++i) s += sh[i];
  return s;
}
__device__ __forceinline__ float block_maxf(float v, float* sh) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) { const float o = __shfl_xor(v, off, 64); v = (v > o || v != v) ? v : o; }
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  float s = sh[0];
  const int nw = blockDim.x >> 6;
  for (int i = 1; i < nw; ++i) { const float o = sh[i]; s = (s > o || s != s) ? s : o; }
  return s;
}

// ---- window moments, three grid-wide passes (double atomics; <= 200k elements) ------------------
// dacc: [0]=sum e  [1]=sum e^2  [2]=sum u  [3]=sum (e-mean_e)^2  [4]=sum (u-mean_u)^2  [5]=sum ac^2 w
// ukey: [0] = key of max |(e-mean)/std|
struct Win {
  int H, W, Hu, Wu, ne, nu;
};
__device__ __forceinline__ Win make_win(const IcsStatsArgs& a) {
  Win w;
  w.H = a.bottom - a.top; w.W = a.right - a.left;
  w.Hu = (a.bottom - a.geo.pad) - (a.top + a.geo.pad); w.Wu = (a.right - a.geo.pad) - (a.left + a.geo.pad);
  if (w.Hu < 0) w.Hu = 0;
  if (w.Wu < 0) w.Wu = 0;
  w.ne = w.H * w.W * 3; w.nu = w.Hu * w.Wu * 3;
  return w;
}
// error window [top:bottom, left:right] in image coordinates -> u-frame (+pad)
__device__ __forceinline__ float win_e(const IcsStatsArgs& a, const Win& w, int i) {
  const int r = i / (3 * w.W), c = i - r * 3 * w.W;
  return a.e[(ptrdiff_t)(a.top + a.geo.pad + r) * a.geo.pitch + 3 * (a.left + a.geo.pad) + c];
}
// u window [top+pad : bottom-pad, left+pad : right-pad] in u coordinates (pyx:600)
__device__ __forceinline__ float win_u(const IcsStatsArgs& a, const Win& w, int i) {
  const int r = i / (3 * w.Wu), c = i - r * 3 * w.Wu;
  return a.u[(ptrdiff_t)(a.top + a.geo.pad + r) * a.geo.pitch + 3 * (a.left + a.geo.pad) + c];
}
__device__ __forceinline__ float mean_of(double s, int n) { return (float)(s / n); }
__device__ __forceinline__ float std_of(double d2, int n) { return sqrtf((float)(d2 / n)); }

__global__ __launch_bounds__(256) void k_mom1(IcsStatsArgs a) {
  __shared__ double shd[4];
  const Win w = make_win(a);
  double s = 0.0, s2 = 0.0, su = 0.0;
  const int stride = gridDim.x * 256, t0 = blockIdx.x * 256 + threadIdx.x;
  for (int i = t0; i < w.ne; i += stride) { const float v = win_e(a, w, i); s += v; s2 += (double)v * v; }
  for (int i = t0; i < w.nu; i += stride) su += win_u(a, w, i);
  s = block_sum(s, shd); s2 = block_sum(s2, shd); su = block_sum(su, shd);
  if (threadIdx.x == 0) { atomicAdd(a.dacc + 0, s); atomicAdd(a.dacc + 1, s2); atomicAdd(a.dacc + 2, su); }
}
// second pass: central moments, and max |e - mean| -- the maximum of |(e - mean) / std| the reference takes (pyx:628-629) is that
// value divided by std (a division by a positive number is monotonic, so the arg-max element gives bit for bit the same quotient);
// as a pass of its own behind this one it was one more 10-us launch per outer iteration
__global__ __launch_bounds__(256) void k_mom2(IcsStatsArgs a) {
  __shared__ double shd[4];
  __shared__ float shf[4];
  const Win w = make_win(a);
  const float mean_e = mean_of(a.dacc[0], w.ne), mean_u = w.nu ? mean_of(a.dacc[2], w.nu) : 0.f;
  double d2 = 0.0, du = 0.0;
  float mx = 0.f;
  const int stride = gridDim.x * 256, t0 = blockIdx.x * 256 + threadIdx.x;
  for (int i = t0; i < w.ne; i += stride) {
    const float v = __fsub_rn(win_e(a, w, i), mean_e); d2 += (double)v * v;
    const float av = __builtin_fabsf(v);
    mx = (mx > av || mx != mx) ? mx : av;
  }
  for (int i = t0; i < w.nu; i += stride) { const float v = __fsub_rn(win_u(a, w, i), mean_u); du += (double)v * v; }
  d2 = block_sum(d2, shd); du = block_sum(du, shd);
  mx = block_maxf(mx, shf);
  if (threadIdx.x == 0) {
    atomicAdd(a.dacc + 3, d2); atomicAdd(a.dacc + 4, du);
    if (a.do_mr) atomicMax(a.ukey, (mx != mx) ? 0xFFC00000u : ics_f2key(mx));
  }
}
// One P-point complex FFT per workgroup (P/2 threads), radix-2 Stockham autosort in LDS; element j of a line lives at
// base + j * stride_elem; forward: exp(-i...), inverse: conjugate, unscaled.  The element-wise passes around the four
// transforms are folded into the loads, and only the lines that matter are transformed:
//   LOAD 1: row transform of the normalised window z = ((e - mean)/std)/max|t| straight from the residual frame, zero
//           padded to P columns; rows >= H of the padded P x P array are zero and are neither written nor transformed
//   LOAD 2: forward column transform; rows >= H are read as zero
//   LOAD 3: inverse column transform of |Z|^2 (Wiener-Khinchin)
//   LOAD 0: inverse row transform of the rows k_mr reads (|row offset| <= H/2, through the line remap)
// line l of a plane is row/column  l < n0 ? l : l + skip.  (As separate kernels -- window fill, four full transforms,
// |Z|^2 -- the same arithmetic took six launches and 0.075 ms.)
struct FftX {
  float2* data; int P, logP; long stride_elem, line_stride; int lines_per_plane; long plane_stride; int inverse; const float2* tw;
  int n0, skip, nvalid;
};
template <int LOAD>
__global__ void k_fftx(FftX f, IcsStatsArgs a) {
  extern __shared__ __attribute__((aligned(16))) float2 sm[];
  float2* x = sm;
  float2* y = sm + f.P;
  const int P = f.P, t = P >> 1, tid = threadIdx.x;
  const int plane = blockIdx.x / f.lines_per_plane;
  int line = blockIdx.x - plane * f.lines_per_plane;
  line = line < f.n0 ? line : line + f.skip;
  float2* base = f.data + (long)plane * f.plane_stride + (long)line * f.line_stride;
  // P / 2 butterflies per stage over min(P / 2, 1024) threads: above P = 2048 (stats windows wider than 1024 px; the reference has
  // no limit, lib/deconvolution.pyx:623-638) a thread takes several
  const int nthr = blockDim.x;
  if (LOAD == 1) {
    const IcsGeom& G = a.geo;
    const Win w = make_win(a);
    const float mean_e = mean_of(a.dacc[0], w.ne), std_e = std_of(a.dacc[3], w.ne);
    const float mxk = ics_key2f(a.ukey[0]);                       // max |e - mean| (k_mom2), NaN key kept
    const float mx = (a.ukey[0] == 0xFFC00000u) ? mxk : __fdiv_rn(mxk, std_e);   // = max |(e - mean) / std|
    for (int j = tid; j < P; j += nthr) {
      float v = 0.f;
      if (line < w.H && j < w.W) {
        const float e = a.e[(ptrdiff_t)(a.top + G.pad + line) * G.pitch + 3 * (a.left + G.pad + j) + plane];
        v = __fdiv_rn(__fdiv_rn(__fsub_rn(e, mean_e), std_e), mx);
      }
      x[j] = make_float2(v, 0.f);
    }
  } else {
    for (int j = tid; j < P; j += nthr) {
      float2 v = make_float2(0.f, 0.f);
      if (LOAD != 2 || j < f.nvalid) v = base[(long)j * f.stride_elem];
      if (LOAD == 3) v = make_float2(v.x * v.x + v.y * v.y, 0.f);
      x[j] = v;
    }
  }
  __syncthreads();
  for (int s = 0, p = 1; s < f.logP; ++s, p <<= 1) {
    for (int b = tid; b < t; b += nthr) {
      const int k = b & (p - 1);
      const int j = ((b - k) << 1) + k;
      float2 w = f.tw[k * (t / p)];
      if (f.inverse) w.y = -w.y;
      const float2 u0 = x[b], v = x[b + t];
      const float2 u1 = make_float2(v.x * w.x - v.y * w.y, v.x * w.y + v.y * w.x);
      y[j] = make_float2(u0.x + u1.x, u0.y + u1.y);
      y[j + p] = make_float2(u0.x - u1.x, u0.y - u1.y);
    }
    __syncthreads();
    float2* tmp = x; x = y; y = tmp;
  }
  for (int j = tid; j < P; j += nthr) base[(long)j * f.stride_elem] = x[j];
}

__device__ void stats_final(const IcsStatsArgs& a);

// sum over (H, W, 3) of ac^2 * w,  ac[a][b] = Z[(a - H/2) mod P][(b - W/2) mod P] / P^2
__global__ __launch_bounds__(256) void k_mr(IcsStatsArgs a) {
  __shared__ double shd[4];
  const int H = a.bottom - a.top, W = a.right - a.left, P = a.P;
  const float inv = 1.0f / ((float)P * (float)P);
  double s = 0.0;
  const int n = H * W * 3;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const int b = i % W, r = (i / W) % H, c = i / (W * H);
    const int zr = (r - H / 2 + P) & (P - 1), zc = (b - W / 2 + P) & (P - 1);
    const float ac = a.z[((long)c * P + zr) * P + zc].x * inv;
    s += (double)__fmul_rn(__fmul_rn(ac, ac), a.weights[r * W + b]);
  }
  s = block_sum(s, shd);
  if (threadIdx.x == 0) {
    atomicAdd(a.dacc + 5, s);
    // the last workgroup out writes the scalars of the outer iteration (one launch less): its own sum is in, and so are all
    // the others' -- every workgroup adds before it takes a ticket (device-scope atomics on the same L2-resident words)
    __threadfence();
    if (atomicAdd(a.ukey + 1, 1u) == gridDim.x - 1) { __threadfence(); stats_final(a); }
  }
}

// scalars of the outer iteration (pyx:593-638) from the accumulators; re-arms the accumulators for the next call
__device__ void stats_final(const IcsStatsArgs& a) {
  const Win w = make_win(a);
  volatile double* dacc = a.dacc;
  a.scal[ICS_SC_HU] = (float)(dacc[1] / ((double)w.H * w.W * 3));
  float varu = __builtin_nanf("");
  if (w.nu > 0) { const float sd = std_of(dacc[4], w.nu); varu = __fmul_rn(sd, sd); }
  a.scal[ICS_SC_VARU] = varu;
  const bool no_dof = a.dofkeys[0] == 0xFFFFFFFFu && a.dofkeys[1] == 0u;   // modes without a DoF blend (PAM)
  a.scal[ICS_SC_DOFMIN] = a.dofkeys[2] ? __builtin_nanf("") : (no_dof ? 0.f : ics_key2f(a.dofkeys[0]));
  a.scal[ICS_SC_DOFMAX] = a.dofkeys[2] ? __builtin_nanf("") : (no_dof ? 0.f : ics_key2f(a.dofkeys[1]));
  if (a.do_mr) a.scal[ICS_SC_MR] = (float)(dacc[5] / w.ne);
#pragma unroll
  for (int i = 0; i < 8; ++i) dacc[i] = 0.0;
  a.ukey[0] = 0u; a.ukey[1] = 0u;
}
__global__ void k_stats_final(IcsStatsArgs a) { stats_final(a); }

__global__ __launch_bounds__(256) void k_hasnan(const float* u, IcsGeom G, int* flag) {
  const int ngx = G.tiles_x * 16;
  const long total = (long)G.uM * ngx;
  int bad = 0;
  for (long gid = (long)blockIdx.x * 256 + threadIdx.x; gid < total; gid += (long)gridDim.x * 256) {
    const int y = (int)(gid / ngx), xp = 4 * (int)(gid - (long)y * ngx);
    const ptrdiff_t o = (ptrdiff_t)y * G.pitch + 3 * xp;
    for (int p = 0; p < 4; ++p)
      if (xp + p < G.uN)
        for (int c = 0; c < 3; ++c) { const float v = u[o + 3 * p + c]; bad |= (v != v); }
  }
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

}  // namespace

hipError_t ics_launch_stats(const IcsStatsArgs& a, hipStream_t s) {
  // (dacc / ukey are zero: at job creation, and re-armed by the previous call's last kernel)
  const int ne = (a.bottom - a.top) * (a.right - a.left) * 3;
  int gb = (ne + 256 * 8 - 1) / (256 * 8); if (gb < 1) gb = 1; if (gb > 512) gb = 512;
  hipLaunchKernelGGL(k_mom1, dim3(gb), dim3(256), 0, s, a);
  hipLaunchKernelGGL(k_mom2, dim3(gb), dim3(256), 0, s, a);
  if (a.do_mr) {
    const int P = a.P;
    const size_t lds = 2 * (size_t)P * sizeof(float2);
    const int nthr = P / 2 < 1024 ? P / 2 : 1024;
    if (lds > 64 * 1024) {   // P = 8192: 128 KB of dynamic LDS (stats windows up to 4096 px)
      static std::atomic<bool> cfg[4][ICS_MAX_DEVICES];
      const int dev = ics_current_device();
      hipError_t e = ics_configure_lds(cfg[0], dev, k_fftx<0>, 128 * 1024);
      if (e == hipSuccess) e = ics_configure_lds(cfg[1], dev, k_fftx<1>, 128 * 1024);
      if (e == hipSuccess) e = ics_configure_lds(cfg[2], dev, k_fftx<2>, 128 * 1024);
      if (e == hipSuccess) e = ics_configure_lds(cfg[3], dev, k_fftx<3>, 128 * 1024);
      if (e != hipSuccess) return e;
    }
    const long plane = (long)P * P;
    const int H = a.bottom - a.top;
    // rows: line = row, elements contiguous; columns: line = column, element stride P
    FftX f = {a.z, P, a.logP, 1L, (long)P, H, plane, 0, a.tw, H, 0, H};
    hipLaunchKernelGGL(k_fftx<1>, dim3(3 * H), dim3(nthr), lds, s, f, a);                     // rows of the window
    f.stride_elem = P; f.line_stride = 1; f.lines_per_plane = P; f.n0 = P;
    hipLaunchKernelGGL(k_fftx<2>, dim3(3 * P), dim3(nthr), lds, s, f, a);                     // columns (rows >= H are zero)
    f.inverse = 1;
    hipLaunchKernelGGL(k_fftx<3>, dim3(3 * P), dim3(nthr), lds, s, f, a);                     // |Z|^2, inverse columns
    // inverse rows: k_mr reads rows (r - H/2) mod P, r < H:  0 .. H - H/2 - 1  and  P - H/2 .. P - 1
    f.stride_elem = 1; f.line_stride = P; f.lines_per_plane = H; f.n0 = H - H / 2; f.skip = P - H;
    hipLaunchKernelGGL(k_fftx<0>, dim3(3 * H), dim3(nthr), lds, s, f, a);
    hipLaunchKernelGGL(k_mr, dim3(gb), dim3(256), 0, s, a);   // its last workgroup writes the scalars
  } else {
    hipLaunchKernelGGL(k_stats_final, dim3(1), dim3(1), 0, s, a);
  }
  return hipGetLastError();
}

hipError_t ics_launch_hasnan(const float* u, const IcsGeom& g, int* flag, hipStream_t s) {
  hipLaunchKernelGGL(k_hasnan, dim3(512), dim3(256), 0, s, u, g, flag);
  return hipGetLastError();
}
