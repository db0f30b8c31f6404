// ics_big.hip -- PSF sizes 65 ... 127: the two convolutions (A1+A2, A3) and the PSF gradient (A12+A13) for any odd size.
//
// The reference has no limit on the PSF size (lib/deconvolution.pyx:341: `MK` is whatever the caller's array is; its convolutions
// are FFTs).  The tuned kernels of this library are compiled per size (matrix cores to 37, packed fp32 to 63); beyond that these
// two kernels take the size at run time.  Plain fp32 FMA streams out of LDS -- built to be correct and to keep the arithmetic units
// busy, not tuned per size: at 4096^2 a 65x65 convolution is 4.2e11 flop per pass.  Under ICS_CONV_AUTO they also serve the upper part
// of the compiled range where they are the faster ones (ics_api.hip, use_big_conv).
//
//   k_conv_big  : out[y, x, c] = sum_{a,b<K} W[a, b, c] * in[y + a - pad, x + b - pad, c]   (u-frame coordinates, as ics_conv.hip)
//                 W = rot180(psf), out = error - image on the M x N interior (mode 0: pyx:477-488)
//                 W = psf,         out = gradu on the whole u-frame          (mode 1: pyx:490-491; the maxima of pyx:523-524 are
//                                                                             taken by k_band_reduce behind it)
//   k_gradk_big : partial[wg][c][a][b] = sum over the workgroup's tiles of E[y, x, c] * U[y + pad - a, x + pad - b, c]
//                 (pyx:567-571), reduced in double by k_gradk_reduce like every other gradient kernel.
#include "ics_kernels.h"

namespace {

constexpr int BIG_TA = 8;                  // kernel rows per staged block
constexpr int BIG_TH = 64, BIG_TW = 64;    // output tile of the convolution (256 threads: two rows x 8 pixels each)
constexpr int BIG_KMAX = 127;
// staged columns per row: LW = 64 + K8 + 8 (run time: the LDS a workgroup takes, and with it the workgroups per CU, follow the PSF size)
constexpr int BIG_RING = BIG_TH + BIG_TA;               // staged rows: a ring, eight new rows per block of kernel rows
static inline int big_lw(int K) { return BIG_TW + ((K + 7) & ~7) + 8; }
static inline size_t big_conv_lds(int K) { return (size_t)BIG_RING * big_lw(K) * sizeof(float) + (size_t)(BIG_TA + 1) * ((K + 7) & ~7) * 2 * sizeof(float); }

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

// in-frame test of a u-frame coordinate (the frames carry an apron of ay rows / ax pixels around the tile grid)
__device__ __forceinline__ bool in_frame(const IcsGeom& G, int y, int x) {
  return y >= -G.ay && y < G.rows - G.ay && x >= -G.ax && 3 * (x + G.ax) + 2 < G.pitch;
}

// One workgroup = a 64 x 64 output tile of one channel at a time; a thread owns two consecutive output rows x 8 pixels, held as
// PAIRS (row A, row B) so that the arithmetic is v_pk_fma_f32 (the fp32 vector peak on CDNA4 needs the packed form: ics_conv.hip):
// an input value of row r meets kernel row a in output row A and kernel row a - 1 in output row B, i.e. one broadcast input
// times the weight pair (W[a][b], W[a-1][b]).  The input rows of the tile sit in an LDS ring of 72 rows: a block of 8 kernel rows
// walks 9 input rows per thread, the next block replaces the 8 oldest rows.  A thread slides a 16-value register window along an
// input row: 2 ds_read_b128 of inputs and 4 of weight pairs per 64 packed FMAs.  Each input row's contribution is summed on its
// own before it enters the total (short rounding chains: the sums run to 16129 terms).
template <int MODE>
__global__ __launch_bounds__(256) void k_conv_big(IcsConvArgs a, const float* __restrict__ psf) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const IcsGeom& G = a.g;
  const int K = G.K, pad = G.pad, K8 = (K + 7) & ~7;
  const int BIG_LW = BIG_TW + K8 + 8;
  float* tile = lds;                                              // [BIG_RING][BIG_LW]
  f2* wp = reinterpret_cast<f2*>(lds + BIG_RING * BIG_LW);        // [BIG_TA + 1][K8] pairs (W[a0 + q][b], W[a0 + q - 1][b])
  const int tid = threadIdx.x, rp = tid >> 3, tc = tid & 7;
  // output region: mode 0 the image interior, mode 1 the whole u-frame
  const int oy0 = MODE == 0 ? pad : 0, ox0 = MODE == 0 ? pad : 0;
  const int oh = MODE == 0 ? G.M : G.uM, ow = MODE == 0 ? G.N : G.uN;
  const int ntx = (ow + BIG_TW - 1) / BIG_TW, nty = (oh + BIG_TH - 1) / BIG_TH;
  for (int t = blockIdx.x; t < ntx * nty; t += gridDim.x) {
    const int y0 = oy0 + (t / ntx) * BIG_TH, x0 = ox0 + (t % ntx) * BIG_TW;
    for (int c = 0; c < 3; ++c) {
      f2 acc[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = (f2){0.f, 0.f};
      for (int a0 = 0; a0 < K; a0 += BIG_TA) {
        const int ta = K - a0 < BIG_TA ? K - a0 : BIG_TA;
        __syncthreads();   // the previous block's readers are done
        // input rows [a0, a0 + 64 + ta) of the tile (row 0 = frame row y0 - pad): all of them for the first block, the new ones after
        const int r_new = a0 == 0 ? 0 : a0 + BIG_TH;            // (ta <= 8 new rows; the first block stages 64 + ta)
        const int n_new = a0 + BIG_TH + ta - r_new;
        for (int i = tid; i < n_new * BIG_LW; i += 256) {
          const int rr = i / BIG_LW, col = i - rr * BIG_LW;
          const int r = r_new + rr;
          const int y = y0 - pad + r, x = x0 - pad + col;
          // (unconditional load from a clamped address, value discarded outside the frame: a conditional load is a branch and a full
          //  wait per element -- the staging ran as a chain of exposed memory latencies)
          const bool ok = in_frame(G, y, x);
          const float val = a.in[ok ? (ptrdiff_t)y * G.pitch + 3 * x + c : (ptrdiff_t)0];
          tile[(r % BIG_RING) * BIG_LW + col] = ok ? val : 0.f;
        }
        for (int i = tid; i < (ta + 1) * K8; i += 256) {
          const int q = i / K8, b = i - q * K8;
          const int bc = b < K ? b : K - 1;
          const int ka = a0 + q < K ? a0 + q : K - 1, kb = a0 + q >= 1 ? a0 + q - 1 : 0;   // clamped: loaded unconditionally, masked below
          const float va = MODE == 0 ? psf[((K - 1 - ka) * K + (K - 1 - bc)) * 3 + c] : psf[(ka * K + bc) * 3 + c];
          const float vb = MODE == 0 ? psf[((K - 1 - kb) * K + (K - 1 - bc)) * 3 + c] : psf[(kb * K + bc) * 3 + c];
          wp[q * K8 + b] = (f2){(b < K && q < ta) ? va : 0.f, (b < K && q >= 1) ? vb : 0.f};
        }
        __syncthreads();
        for (int q = 0; q <= ta; ++q) {
          const int r = (a0 + 2 * rp + q) % BIG_RING;
          const float* row = tile + r * BIG_LW + 8 * tc;
          const f2* wr = wp + q * K8;
          f2 r8[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) r8[i] = (f2){0.f, 0.f};
          float v[16];
          { const f4 p = *reinterpret_cast<const f4*>(row), s4 = *reinterpret_cast<const f4*>(row + 4);
            v[8] = p.x; v[9] = p.y; v[10] = p.z; v[11] = p.w; v[12] = s4.x; v[13] = s4.y; v[14] = s4.z; v[15] = s4.w; }
          for (int b0 = 0; b0 < K8; b0 += 8) {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = v[8 + k];
            { const f4 p = *reinterpret_cast<const f4*>(row + b0 + 8), s4 = *reinterpret_cast<const f4*>(row + b0 + 12);
              v[8] = p.x; v[9] = p.y; v[10] = p.z; v[11] = p.w; v[12] = s4.x; v[13] = s4.y; v[14] = s4.z; v[15] = s4.w; }
            f2 w[8];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const f4 ww = *reinterpret_cast<const f4*>(wr + b0 + 2 * k);
              w[2 * k] = (f2){ww.x, ww.y}; w[2 * k + 1] = (f2){ww.z, ww.w};
            }
#pragma unroll
            for (int bb = 0; bb < 8; ++bb)
#pragma unroll
              for (int i = 0; i < 8; ++i) r8[i] = __builtin_elementwise_fma(w[bb], (f2){v[bb + i], v[bb + i]}, r8[i]);
          }
#pragma unroll
          for (int i = 0; i < 8; ++i) acc[i] += r8[i];
        }
      }
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int y = y0 + 2 * rp + h;
        if (y < oy0 + oh) {
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const int x = x0 + 8 * tc + i;
            if (x < ox0 + ow) {
              const ptrdiff_t o = (ptrdiff_t)y * G.pitch + 3 * x + c;
              const float r = h ? acc[i].y : acc[i].x;
              a.out[o] = MODE == 0 ? __fsub_rn(r, a.f[o]) : r;
            }
          }
        }
      }
    }
  }
}

// ---- PSF gradient -----------------------------------------------------------------------------------------------------------
constexpr int GB_T = 32;   // residual tile: 32 x 32 pixels of one channel
struct GradkBig { int K, LWU, ntask, nchunk; };

__global__ __launch_bounds__(256) void k_gradk_big(IcsGradkArgs a, GradkBig cfg) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const IcsGeom& G = a.geo;
  const int K = cfg.K, pad = G.pad, LWU = cfg.LWU, NT = 16 * ((K + 15) / 16);
  const int UR = GB_T + K - 1;               // staged rows of u
  float* ul = lds;                           // [UR][LWU], column 0 = u column x0 + pad - (K - 1) - 8
  float* el = lds + (size_t)UR * LWU;        // [GB_T][GB_T]
  const int tid = threadIdx.x;
  // residual tiles over the image interior (outside it the residual is zero)
  const int ntx = (G.N + GB_T - 1) / GB_T, nty = (G.M + GB_T - 1) / GB_T, ntile = ntx * nty;
  const int per = (ntile + gridDim.x - 1) / gridDim.x;
  const int t_begin = blockIdx.x * per, t_end = t_begin + per < ntile ? t_begin + per : ntile;
  float* dst = a.partial + (size_t)blockIdx.x * (3 * NT * NT);
  // a thread's tasks: task = chunk * K + ka (consecutive lanes = consecutive kernel rows: LWU is odd, so their LDS rows hit
  // different banks), eight taps b = 8 chunk ... 8 chunk + 7 each
  constexpr int MAXQ = (BIG_KMAX * 16 + 255) / 256;   // 8
  // Accumulators restart with every tile and are folded into the workgroup's partial block (plain read-modify-write: the block is
  // this workgroup's own): one fp32 chain over all tiles of a workgroup -- 65 k terms at 4096^2 -- rounds ~5x worse than 1024-term
  // chains whose sums are then added (1.4e-5 against the 1e-5 gate when few workgroups walk many tiles).
  for (int c = 0; c < 3; ++c) {
    if (t_begin >= t_end) {   // no tile for this workgroup: its partial block is zero
#pragma unroll
      for (int q = 0; q < MAXQ; ++q) {
        const int task = tid + 256 * q;
        if (task < cfg.ntask) {
          const int chunk = task / K, ka = task - chunk * K;
#pragma unroll
          for (int i = 0; i < 8; ++i) if (8 * chunk + i < K) dst[((size_t)c * NT + ka) * NT + 8 * chunk + i] = 0.f;
        }
      }
    }
    for (int t = t_begin; t < t_end; ++t) {
      float acc[MAXQ][8];
#pragma unroll
      for (int q = 0; q < MAXQ; ++q)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[q][i] = 0.f;
      const int y0 = pad + (t / ntx) * GB_T, x0 = pad + (t % ntx) * GB_T;
      __syncthreads();
      for (int i = tid; i < UR * LWU; i += 256) {
        const int r = i / LWU, col = i - r * LWU;
        const int y = y0 + pad - (K - 1) + r, x = x0 + pad - (K - 1) - 8 + col;
        const bool ok = in_frame(G, y, x);
        const float val = a.u[ok ? (ptrdiff_t)y * G.pitch + 3 * x + c : (ptrdiff_t)0];
        ul[i] = ok ? val : 0.f;
      }
      for (int i = tid; i < GB_T * GB_T; i += 256) {
        const int r = i / GB_T, col = i - r * GB_T;
        const int y = y0 + r, x = x0 + col;
        const bool ok = y < pad + G.M && x < pad + G.N;
        const float val = a.e[ok ? (ptrdiff_t)y * G.pitch + 3 * x + c : (ptrdiff_t)0];
        el[i] = ok ? val : 0.f;
      }
      __syncthreads();
#pragma unroll
      for (int q = 0; q < MAXQ; ++q) {
        const int task = tid + 256 * q;
        if (task < cfg.ntask) {
          const int chunk = task / K, ka = task - chunk * K;
          // tap (ka, b) at residual pixel (ey, ex) reads u_l[ey + K - 1 - ka][8 + ex + K - 1 - b]
          const float* ubase = ul + (size_t)(K - 1 - ka) * LWU + 8 + (K - 1 - 8 * chunk);
          for (int ey = 0; ey < GB_T; ++ey) {
            const float* urow = ubase + (size_t)ey * LWU;   // urow[ex - i] for tap b = 8 chunk + i
            const float* erow = el + ey * GB_T;
            float v[16];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[8 + k] = urow[k - 8];
            for (int ex0 = 0; ex0 < GB_T; ex0 += 8) {
#pragma unroll
              for (int k = 0; k < 8; ++k) v[k] = v[8 + k];
#pragma unroll
              for (int k = 0; k < 8; ++k) v[8 + k] = urow[ex0 + k];
              const f4 e0 = *reinterpret_cast<const f4*>(erow + ex0), e1 = *reinterpret_cast<const f4*>(erow + ex0 + 4);
              const float e[8] = {e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w};
#pragma unroll
              for (int k = 0; k < 8; ++k)
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[q][i] = __builtin_fmaf(e[k], v[8 + k - i], acc[q][i]);
            }
          }
          const bool first = t == t_begin;
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const int b = 8 * chunk + i;
            if (b < K) {
              float* d = dst + ((size_t)c * NT + ka) * NT + b;
              *d = first ? acc[q][i] : *d + acc[q][i];
            }
          }
        }
      }
    }
  }
}

// ---- tap blocks on the matrix cores (PSF sizes 51 ... 127) ---------------------------------------------------------------------
// A convolution is linear in its taps: the K x K PSF is cut into nblk x nblk blocks of Kb x Kb taps (Kb odd, 23 ... 33: sizes the
// matrix-core convolution is built for), block (qa, qb) is convolved by k_conv_mfma<Kb> with the input pointer shifted by
// (qa Kb + Kb/2 - pad, qb Kb + Kb/2 - pad), and the block results are added (ics_api.hip, do_conv_blocks).  This kernel packs the
// block weight tables in the format k_psf packs the whole-PSF tables in (ics_common.h): one workgroup per block, both orientations.
__global__ __launch_bounds__(256) void k_pack_blocks(const float* __restrict__ psf, int K, int Kb, int nblk, void* tconv, void* tcorr, size_t table_floats) {
  __shared__ uint32_t smax;
  const int tid = threadIdx.x, qa = blockIdx.x / nblk, qb = blockIdx.x - qa * nblk, a0 = qa * Kb, b0 = qb * Kb;
  if (tid == 0) smax = 0u;
  __syncthreads();
  uint32_t km = 0u;
  for (int i = tid; i < 3 * K * K; i += 256) { const uint32_t k1 = __float_as_uint(__builtin_fabsf(psf[i])); km = km > k1 ? km : k1; }   // (bit patterns: NaN on top)
  km = ics_wave_max_u32(km);
  if ((tid & 63) == 0) atomicMax(&smax, km);
  __syncthreads();
  const float m = __uint_as_float(smax);
  const uint32_t e = (smax >> 23) & 0xFFu;
  uint32_t sb = 127u;
  if (m > 0.f && e != 255u) { sb = 268u - e; sb = sb > 240u ? 240u : sb; }
  const float s_w = __uint_as_float(sb << 23), inv_w = __uint_as_float((254u - sb) << 23);
  // the residual's blocks are chained with alternating signs, the last one positive (ics_api.hip, do_conv_blocks)
  const float s_wc = ((nblk * nblk - 1 - (int)blockIdx.x) & 1) ? -s_w : s_w;
  _Float16* tc = reinterpret_cast<_Float16*>(reinterpret_cast<float*>(tconv) + (size_t)blockIdx.x * table_floats);
  _Float16* tr = reinterpret_cast<_Float16*>(reinterpret_cast<float*>(tcorr) + (size_t)blockIdx.x * table_floats);
  const int rh = ((2 * (Kb + 17) + 3) & ~3) / 2;      // halves per row (MCfg::WROWB / 2)
  const int nhalf = 3 * Kb * 2 * rh;
  for (int i = tid; i < nhalf; i += 256) {
    const int ent = i / rh, hh = i - ent * rh;
    const int sp = ent & 1, ca = ent >> 1, c = ca / Kb, ra = ca - c * Kb;
    const int b = hh - 7, A = a0 + ra, B = b0 + b;
    float w1 = 0.f, w2 = 0.f;
    if (b >= 0 && b < Kb && A < K && B < K) {
      w1 = psf[(A * K + B) * 3 + c] * s_w;                              // correlation orientation (A3): W = psf
      w2 = psf[((K - 1 - A) * K + (K - 1 - B)) * 3 + c] * s_wc;         // convolution orientation (A1): W = rot180(psf), with the block's sign
    }
    const _Float16 h1 = (_Float16)w1, h2 = (_Float16)w2;
    const int o = ca * 2 * rh + 4 * (hh >> 1) + 2 * sp + (hh & 1);
    tr[o] = sp ? (_Float16)(w1 - (float)h1) : h1;
    tc[o] = sp ? (_Float16)(w2 - (float)h2) : h2;
  }
  if (tid == 0) {
    *reinterpret_cast<float*>(tc + nhalf) = inv_w;
    *reinterpret_cast<float*>(tr + nhalf) = inv_w;
  }
}

// out[y][f] += add[y][f] on rows [y0, y1), floats [f0, f1) of a frame row (origin-relative)
__global__ __launch_bounds__(256) void k_frame_add(float* __restrict__ out, const float* __restrict__ add, int pitch, int y0, int y1, int f0, int f1) {
  const int w = f1 - f0;
  const long total = (long)(y1 - y0) * w;
  for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
    const int r = (int)(t / w), f = (int)(t - (long)r * w);
    const ptrdiff_t o = (ptrdiff_t)(y0 + r) * pitch + f0 + f;
    out[o] = __fadd_rn(out[o], add[o]);
  }
}

__global__ __launch_bounds__(256) void k_frame_neg(float* __restrict__ out, const float* __restrict__ in, size_t count) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256) out[i] = -in[i];
}

}  // namespace

hipError_t ics_launch_frame_neg(float* out, const float* in, size_t count, hipStream_t s) {
  hipLaunchKernelGGL(k_frame_neg, dim3(2048), dim3(256), 0, s, out, in, count);
  return hipGetLastError();
}
hipError_t ics_launch_pack_blocks(const float* psf, int K, int Kb, int nblk, void* tconv, void* tcorr, size_t table_floats, hipStream_t s) {
  hipLaunchKernelGGL(k_pack_blocks, dim3(nblk * nblk), dim3(256), 0, s, psf, K, Kb, nblk, tconv, tcorr, table_floats);
  return hipGetLastError();
}
hipError_t ics_launch_frame_add(float* out, const float* add, int pitch, int y0, int y1, int f0, int f1, hipStream_t s) {
  hipLaunchKernelGGL(k_frame_add, dim3(2048), dim3(256), 0, s, out, add, pitch, y0, y1, f0, f1);
  return hipGetLastError();
}

bool ics_big_supported(int K) { return K > 63 && K <= BIG_KMAX && (K & 1); }

hipError_t ics_launch_conv_big(int mode, const IcsConvArgs& a, const float* psf, hipStream_t s) {
  if (mode != 0 && mode != 1) return hipErrorInvalidValue;
  static std::atomic<bool> cfg[2][ICS_MAX_DEVICES];
  const int dev = ics_current_device();
  hipError_t e = ics_configure_lds(cfg[0], dev, k_conv_big<0>, big_conv_lds(BIG_KMAX));
  if (e == hipSuccess) e = ics_configure_lds(cfg[1], dev, k_conv_big<1>, big_conv_lds(BIG_KMAX));
  if (e != hipSuccess) return e;
  const size_t lds = big_conv_lds(a.g.K);
  int per_cu = (int)((160 * 1024) / lds); per_cu = per_cu > 4 ? 4 : (per_cu < 1 ? 1 : per_cu);   // 108 VGPRs: four workgroups of four waves
  const int grid = per_cu * ics_device_cus(dev);
  if (mode == 0) hipLaunchKernelGGL(k_conv_big<0>, dim3(grid), dim3(256), lds, s, a, psf);
  else hipLaunchKernelGGL(k_conv_big<1>, dim3(grid), dim3(256), lds, s, a, psf);
  return hipGetLastError();
}

hipError_t ics_launch_gradk_big(const IcsGradkArgs& a, int nblocks, hipStream_t s) {
  const int K = a.geo.K;
  GradkBig cfg;
  cfg.K = K; cfg.nchunk = (K + 7) / 8; cfg.ntask = K * cfg.nchunk;
  cfg.LWU = (8 + GB_T + K - 1 + 8) | 1;   // odd: consecutive rows start in consecutive banks
  const size_t lds = ((size_t)(GB_T + K - 1) * cfg.LWU + GB_T * GB_T) * sizeof(float);
  static std::atomic<bool> done[ICS_MAX_DEVICES];
  hipError_t e = ics_configure_lds(done, ics_current_device(), k_gradk_big, 160 * 1024);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_gradk_big, dim3(nblocks), dim3(256), lds, s, a, cfg);
  return hipGetLastError();
}
