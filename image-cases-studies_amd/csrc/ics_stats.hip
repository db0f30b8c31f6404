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

// A thread takes 8 elements per trip and requests them together: as a plain loop (index arithmetic with two integer divisions, then
// one load, then the add) every element exposed the full memory latency -- 8 x ~1 us of these passes' 12 and 9 us.  The index of
// an element past the end is clamped and its value discarded: a conditional load compiles to a branch and a full wait per element.
#define ICS_MOM_BATCH 8
__global__ __launch_bounds__(256) void k_mom1(IcsStatsArgs a) {
  __shared__ double shd[4];
  const Win w = make_win(a);
  double s = 0.0, s2 = 0.0, su = 0.0;
  const int stride = gridDim.x * 256, t0 = blockIdx.x * 256 + threadIdx.x;
  for (int i0 = t0; i0 < w.ne; i0 += ICS_MOM_BATCH * stride) {
    float v[ICS_MOM_BATCH];
#pragma unroll
    for (int k = 0; k < ICS_MOM_BATCH; ++k) { const int i = i0 + k * stride; const float t = win_e(a, w, i < w.ne ? i : w.ne - 1); v[k] = i < w.ne ? t : 0.f; }
#pragma unroll
    for (int k = 0; k < ICS_MOM_BATCH; ++k) { s += v[k]; s2 += (double)v[k] * v[k]; }
  }
  for (int i0 = t0; i0 < w.nu; i0 += ICS_MOM_BATCH * stride) {
    float v[ICS_MOM_BATCH];
#pragma unroll
    for (int k = 0; k < ICS_MOM_BATCH; ++k) { const int i = i0 + k * stride; const float t = win_u(a, w, i < w.nu ? i : w.nu - 1); v[k] = i < w.nu ? t : 0.f; }
#pragma unroll
    for (int k = 0; k < ICS_MOM_BATCH; ++k) su += v[k];
  }
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
  for (int i0 = t0; i0 < w.ne; i0 += ICS_MOM_BATCH * stride) {
    float e[ICS_MOM_BATCH];
#pragma unroll
    for (int k = 0; k < ICS_MOM_BATCH; ++k) { const int i = i0 + k * stride; e[k] = win_e(a, w, i < w.ne ? i : w.ne - 1); }
#pragma unroll
    for (int k = 0; k < ICS_MOM_BATCH; ++k) {
      const float v = i0 + k * stride < w.ne ? __fsub_rn(e[k], mean_e) : 0.f; d2 += (double)v * v;
      const float av = __builtin_fabsf(v);
      mx = (mx > av || mx != mx) ? mx : av;
    }
  }
  for (int i0 = t0; i0 < w.nu; i0 += ICS_MOM_BATCH * stride) {
    float e[ICS_MOM_BATCH];
#pragma unroll
    for (int k = 0; k < ICS_MOM_BATCH; ++k) { const int i = i0 + k * stride; e[k] = win_u(a, w, i < w.nu ? i : w.nu - 1); }
#pragma unroll
    for (int k = 0; k < ICS_MOM_BATCH; ++k) { const float v = i0 + k * stride < w.nu ? __fsub_rn(e[k], mean_u) : 0.f; du += (double)v * v; }
  }
  d2 = block_sum(d2, shd); du = block_sum(du, shd);
  mx = block_maxf(mx, shf);
  if (threadIdx.x == 0) {
    atomicAdd(a.dacc + 3, d2); atomicAdd(a.dacc + 4, du);
    if (a.do_mr) atomicMax(a.ukey, (mx != mx) ? 0xFFC00000u : ics_f2key(mx));
  }
}
// Radix-2 Stockham autosort FFTs in LDS.  C lines of P points are transformed together (line c at x + c * LP); P / 2 butterflies per
// line and stage over the workgroup's threads; forward: exp(-i...), inverse: conjugate, unscaled.  Returns the buffer holding the
// result.  The element-wise passes around the four transforms are folded into three kernels, and only the lines that matter are
// transformed:
//   k_fft_rows : row transform of the normalised window z = ((e - mean)/std)/max|t| straight from the residual frame, zero padded to
//                P columns; rows >= H of the padded P x P array are zero and are neither written nor transformed
//   k_fft_cols : forward column transform (rows >= H read as zero), |Z|^2 (Wiener-Khinchin) and the inverse column transform with
//                the column staying in LDS; a workgroup takes C adjacent columns (32-byte segments of each row instead of 8-byte
//                ones) and writes back only the rows the last pass reads
//   k_fft_mr   : inverse row transform of the rows with |row offset| <= H/2 and, from LDS, this row's part of
//                sum ac^2 w  (pyx:633-638); the last workgroup out writes the scalars of the outer iteration
// (Round 2 ran this as rows, columns, |Z|^2 + inverse columns, inverse rows, and a separate weighted sum: five launches and two more
//  round trips of the 6 MB spectrum through L2 / HBM; as six separate element-wise + transform kernels before that.)
__device__ __forceinline__ float2* fft_lines(float2* x, float2* y, int P, int logP, int C, int LP, const float2* __restrict__ tw, bool inverse) {
  const int t = P >> 1, tid = threadIdx.x, nthr = blockDim.x, logt = logP - 1;
  for (int s = 0, p = 1; s < logP; ++s, p <<= 1) {
    for (int q = tid; q < C * t; q += nthr) {
      const int cc = q >> logt, b = q & (t - 1);
      const int k = b & (p - 1);
      const int j = ((b - k) << 1) + k;
      float2 w = tw[k * (t / p)];
      if (inverse) w.y = -w.y;
      const float2* xl = x + cc * LP; float2* yl = y + cc * LP;
      const float2 u0 = xl[b], v = xl[b + t];
      const float2 u1 = make_float2(v.x * w.x - v.y * w.y, v.x * w.y + v.y * w.x);
      yl[j] = make_float2(u0.x + u1.x, u0.y + u1.y);
      yl[j + p] = make_float2(u0.x - u1.x, u0.y - u1.y);
    }
    __syncthreads();
    float2* tmp = x; x = y; y = tmp;
  }
  return x;
}

__global__ void k_fft_rows(IcsStatsArgs a, int C, int LP) {   // grid ceil(3 * H / C), min(C * P / 2, 1024) threads
  extern __shared__ __attribute__((aligned(16))) float2 sm[];
  const IcsGeom& G = a.geo;
  const Win w = make_win(a);
  const int P = a.P, tid = threadIdx.x, nthr = blockDim.x, NL = 3 * w.H;
  const float mean_e = mean_of(a.dacc[0], w.ne), std_e = std_of(a.dacc[3], w.ne);
  const float mxk = ics_key2f(a.ukey[0]);                       // max |e - mean| (k_mom2), NaN key kept
  const float mx = (a.ukey[0] == 0xFFC00000u) ? mxk : __fdiv_rn(mxk, std_e);   // = max |(e - mean) / std|
  for (int i = tid; i < C * P; i += nthr) {
    const int cc = i >> a.logP, j = i & (P - 1), id = blockIdx.x * C + cc;
    float v = 0.f;
    if (id < NL && j < w.W) {
      const int plane = id / w.H, line = id - plane * w.H;
      const float e = a.e[(ptrdiff_t)(a.top + G.pad + line) * G.pitch + 3 * (a.left + G.pad + j) + plane];
      v = __fdiv_rn(__fdiv_rn(__fsub_rn(e, mean_e), std_e), mx);
    }
    sm[cc * LP + j] = make_float2(v, 0.f);
  }
  __syncthreads();
  const float2* r = fft_lines(sm, sm + C * LP, P, a.logP, C, LP, a.tw, false);
  for (int i = tid; i < C * P; i += nthr) {
    const int cc = i >> a.logP, j = i & (P - 1), id = blockIdx.x * C + cc;
    if (id < NL) {
      const int plane = id / w.H, line = id - plane * w.H;
      a.z[((long)plane * P + line) * P + j] = r[cc * LP + j];
    }
  }
}

__global__ void k_fft_cols(IcsStatsArgs a, int C, int LP) {   // grid 3 * P / C, min(C * P / 2, 1024) threads
  extern __shared__ __attribute__((aligned(16))) float2 sm[];
  const int P = a.P, tid = threadIdx.x, nthr = blockDim.x;
  const int H = a.bottom - a.top, n0 = H - H / 2;
  const int groups = P / C;
  const int plane = blockIdx.x / groups, c0 = (blockIdx.x - plane * groups) * C;
  float2* base = a.z + (long)plane * P * P + c0;
  float2* x = sm;
  float2* y = sm + C * LP;
  for (int i = tid; i < C * P; i += nthr) {
    const int cc = i & (C - 1), j = i / C;
    x[cc * LP + j] = j < H ? base[(long)j * P + cc] : make_float2(0.f, 0.f);
  }
  __syncthreads();
  float2* r = fft_lines(x, y, P, a.logP, C, LP, a.tw, false);
  for (int i = tid; i < C * P; i += nthr) {
    const int cc = i / P, j = i - cc * P;
    const float2 v = r[cc * LP + j];
    r[cc * LP + j] = make_float2(v.x * v.x + v.y * v.y, 0.f);
  }
  __syncthreads();
  const float2* q = fft_lines(r, r == x ? y : x, P, a.logP, C, LP, a.tw, true);
  for (int i = tid; i < C * P; i += nthr) {
    const int cc = i & (C - 1), j = i / C;
    if (j < n0 || j >= P - H / 2) base[(long)j * P + cc] = q[cc * LP + j];   // rows (r - H/2) mod P, r < H
  }
}

__device__ void stats_final(const IcsStatsArgs& a);

// sum over (H, W, 3) of ac^2 * w,  ac[r][b] = Z[(r - H/2) mod P][(b - W/2) mod P] / P^2: C rows (of any channel) per workgroup
__global__ void k_fft_mr(IcsStatsArgs a, int C, int LP) {   // grid ceil(3 * H / C), min(C * P / 2, 1024) threads
  extern __shared__ __attribute__((aligned(16))) float2 sm[];
  __shared__ double shd[16];
  const int P = a.P, tid = threadIdx.x, nthr = blockDim.x;
  const int H = a.bottom - a.top, W = a.right - a.left, n0 = H - H / 2, NL = 3 * H;
  for (int i = tid; i < C * P; i += nthr) {
    const int cc = i >> a.logP, j = i & (P - 1), id = blockIdx.x * C + cc;
    float2 v = make_float2(0.f, 0.f);
    if (id < NL) {
      const int plane = id / H, l = id - plane * H;
      const int zr = l < n0 ? l : l + (P - H);
      v = a.z[((long)plane * P + zr) * P + j];
    }
    sm[cc * LP + j] = v;
  }
  __syncthreads();
  const float2* x = fft_lines(sm, sm + C * LP, P, a.logP, C, LP, a.tw, true);
  const float inv = 1.0f / ((float)P * (float)P);
  double s = 0.0;
  for (int i = tid; i < C * W; i += nthr) {
    const int cc = i / W, b = i - cc * W, id = blockIdx.x * C + cc;
    if (id < NL) {
      const int l = id % H;
      const int r = l < n0 ? l + H / 2 : l - n0;
      const int zc = (b - W / 2 + P) & (P - 1);
      const float ac = x[cc * LP + zc].x * inv;
      s += (double)__fmul_rn(__fmul_rn(ac, ac), a.weights[r * W + b]);
    }
  }
  s = block_sum(s, shd);
  if (tid == 0) {
    atomicAdd(a.dacc + 5, s);
    // the last workgroup out writes the scalars of the outer iteration: its own sum is in, and so are all the others' -- every
    // workgroup adds before it takes a ticket (device-scope atomics on the same L2-resident words)
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
  if (a.rearm) {   // what ics_rl_run otherwise queues per outer iteration as a memset and a 16-byte upload (two stream operations, ~15 us)
    a.dofkeys[0] = 0xFFFFFFFFu; a.dofkeys[1] = 0u; a.dofkeys[2] = 0u; a.dofkeys[3] = 0u;
    if (a.red) for (int i = 0; i < 8 * ICS_RED_STRIDE; ++i) a.red[i] = 0u;
  }
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
  int gb = (ne + 256 * ICS_MOM_BATCH - 1) / (256 * ICS_MOM_BATCH); if (gb < 1) gb = 1; if (gb > 512) gb = 512;
  hipLaunchKernelGGL(k_mom1, dim3(gb), dim3(256), 0, s, a);
  hipLaunchKernelGGL(k_mom2, dim3(gb), dim3(256), 0, s, a);
  if (a.do_mr) {
    const int P = a.P, H = a.bottom - a.top;
    const int C = P < 4 ? 1 : (P <= 1024 ? 4 : (P <= 2048 ? 2 : 1));   // columns per workgroup (LDS: 2 * C * LP * 8 bytes); P >= 2
    const int LP = P + (C > 1 ? 4 : 0);                       // line pitch in LDS: the transposing accesses spread over the banks
    const size_t lds = 2 * (size_t)C * LP * sizeof(float2);
    auto clampt = [](int v) { return v < 64 ? 64 : (v > 1024 ? 1024 : v); };   // whole waves (block_sum), at most 1024 threads
    const int nthr = clampt(C * (P / 2));
    if (lds > 64 * 1024) {   // P >= 1024: up to 128 KB of dynamic LDS (stats windows up to 4096 px)
      static std::atomic<bool> cfg[3][ICS_MAX_DEVICES];
      const int dev = ics_current_device();
      hipError_t e = ics_configure_lds(cfg[0], dev, k_fft_rows, 132 * 1024);
      if (e == hipSuccess) e = ics_configure_lds(cfg[1], dev, k_fft_cols, 132 * 1024);
      if (e == hipSuccess) e = ics_configure_lds(cfg[2], dev, k_fft_mr, 132 * 1024);
      if (e != hipSuccess) return e;
    }
    const int glines = (3 * H + C - 1) / C;
    // (the forward rows run one line per workgroup: 8.3 us against 10.0 with four -- their loads are the strided ones)
    hipLaunchKernelGGL(k_fft_rows, dim3(3 * H), dim3(clampt(P / 2)), 2 * (size_t)P * sizeof(float2), s, a, 1, P);
    hipLaunchKernelGGL(k_fft_cols, dim3(3 * (P / C)), dim3(nthr), lds, s, a, C, LP);
    hipLaunchKernelGGL(k_fft_mr, dim3(glines), dim3(nthr), lds, s, a, C, LP);   // its last workgroup writes the scalars
  } else {
    hipLaunchKernelGGL(k_stats_final, dim3(1), dim3(1), 0, s, a);
  }
  return hipGetLastError();
}

hipError_t ics_launch_hasnan(const float* u, const IcsGeom& g, int* flag, hipStream_t s) {
  hipLaunchKernelGGL(k_hasnan, dim3(512), dim3(256), 0, s, u, g, flag);
  return hipGetLastError();
}
