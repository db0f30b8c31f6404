// ics_conv.hip -- the two PSF convolutions of one Richardson-Lucy inner iteration, gfx950.
//
//   mode 0 (A1+A2, lib/deconvolution.pyx:477-488):  error = convolve(u, psf, "valid") - image
//   mode 1 (A3,    lib/deconvolution.pyx:490-491):  gradu = convolve(error, rot180(psf), "full")
//           fused with the reductions of A7 (pyx:523-524): per channel max|lambd*gradu+(u-ut)/2|
//           and max u, so the step size never leaves the device.
//
// In u-frame coordinates (ics_common.h) both are the same symmetric-window correlation
//     out[y, x, c] = sum_{a,b<K} W[a, b, c] * in[y + a - pad, x + b - pad, c]
// with W = rot180(psf) for mode 0 and W = psf for mode 1 (SURVEY.md 8a "exact index forms"); the
// zero apron of the frames supplies the zero extension that `full` needs.
//
// Kernel shape (CDNA4): one 256-thread workgroup (4 waves) per 64 x (16*R) pixel tile.  The tile
// plus halo is staged once in LDS (dwordx4 global loads, HWC rows are contiguous so a 64-px row
// segment is one 768-B run).  Each lane owns R output rows x 4 pixels (12 floats of the flattened
// x*3+c axis, so the channel of a register is a compile-time constant) and walks the input rows
// once: an LDS row strip is read with ds_read_b128 into registers and feeds all R output rows
// (kernel row a = i - r), i.e. K*12 FMAs per output row per strip.  PSF weights are wave-uniform
// and come through the scalar cache into SGPRs (v_fmac_f32 with an SGPR operand), so the VALU
// stream is almost pure FMA.  fp32 throughout; the dense formulation needed for MFMA would waste
// >= 50 % of the matrix pipe on the Toeplitz band, so this path is VALU by design.
#include "ics_common.h"

namespace {

template <int K, int R>
struct ConvCfg {
  static constexpr int PAD = K / 2;
  static constexpr int AX = (PAD + 3) & ~3;
  static constexpr int TW = ICS_TILE;
  static constexpr int TH = 16 * R;
  static constexpr int LROWS = TH + K - 1;
  static constexpr int LW_USED = 3 * (TW + 2 * AX);  // floats staged per LDS row (multiple of 4)
  static constexpr int LALIGN = 64 / R;               // R*LWF % 64 == 0 keeps ds_read_b128 conflict-free
  static constexpr int LWF = ((LW_USED + LALIGN - 1) / LALIGN) * LALIGN;
  static constexpr int OFF0 = 3 * (AX - PAD);
  static constexpr int STRIP = (OFF0 + 12 + 3 * (K - 1) + 3) & ~3;
  static constexpr int WROW = (3 * K + 3) & ~3;
  static constexpr size_t LDS_BYTES = (size_t)LROWS * LWF * 4;
  static_assert(12 * 15 + STRIP <= LWF, "strip overruns the LDS row");
  static_assert(ICS_TILE % TH == 0, "tile height must divide the frame granularity");
};

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    uint32_t o = (uint32_t)__shfl_xor((int)v, off, 64);
    v = v > o ? v : o;
  }
  return v;
}

__device__ __forceinline__ uint32_t key_of(float f) {
  // canonical positive NaN so that a NaN propagates through the integer max like np.amax does
  if (f != f) return 0xFFC00000u;
  return ics_f2key(f);
}

template <int K, int R, int MODE>
__global__ __launch_bounds__(256) void k_conv(IcsConvArgs a) {
  using C = ConvCfg<K, R>;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x;
  const int tx = tid & 15, ty = tid >> 4;
  const int txi = blockIdx.x, tyi = blockIdx.y;
  const int x0 = txi * C::TW, y0 = tyi * C::TH;
  const int pitch = a.g.pitch;

  // ---- stage tile + halo -------------------------------------------------------------------
  {
    const float* src = a.in + (ptrdiff_t)(y0 - C::PAD) * pitch + 3 * (x0 - C::AX);
    constexpr int LW4 = C::LW_USED / 4;
    for (int v = tid; v < C::LROWS * LW4; v += 256) {
      const int row = v / LW4, c4 = v - row * LW4;
      const float4 val = *reinterpret_cast<const float4*>(src + (ptrdiff_t)row * pitch + 4 * c4);
      *reinterpret_cast<float4*>(lds + row * C::LWF + 4 * c4) = val;
    }
  }
  __syncthreads();

  float acc[R][12];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int f = 0; f < 12; ++f) acc[r][f] = 0.f;

  const float* lrow0 = lds + (ty * R) * C::LWF + 12 * tx;
  const float* __restrict__ wbase = a.w;

#pragma unroll 1
  for (int i = 0; i < R + K - 1; ++i) {
    float strip[C::STRIP];
    const float4* lp = reinterpret_cast<const float4*>(lrow0 + i * C::LWF);
#pragma unroll
    for (int j = 0; j < C::STRIP / 4; ++j) {
      const float4 t = lp[j];
      strip[4 * j + 0] = t.x; strip[4 * j + 1] = t.y; strip[4 * j + 2] = t.z; strip[4 * j + 3] = t.w;
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int arow = i - r;  // wave-uniform kernel row feeding output row r from input row i
      if (arow >= 0 && arow < K) {
        const float* __restrict__ wr = wbase + arow * C::WROW;
#pragma unroll
        for (int b = 0; b < K; ++b) {
          const float w0 = wr[3 * b + 0], w1 = wr[3 * b + 1], w2 = wr[3 * b + 2];
#pragma unroll
          for (int p = 0; p < 4; ++p) {
            acc[r][3 * p + 0] = __builtin_fmaf(w0, strip[C::OFF0 + 3 * (p + b) + 0], acc[r][3 * p + 0]);
            acc[r][3 * p + 1] = __builtin_fmaf(w1, strip[C::OFF0 + 3 * (p + b) + 1], acc[r][3 * p + 1]);
            acc[r][3 * p + 2] = __builtin_fmaf(w2, strip[C::OFF0 + 3 * (p + b) + 2], acc[r][3 * p + 2]);
          }
        }
      }
    }
  }

  // ---- epilogue ----------------------------------------------------------------------------
  const int xp = x0 + 4 * tx;  // first of this lane's 4 pixels (u-frame x)
  if (MODE == 0) {
    // error = synth - image on the M x N interior (pyx:488); the border ring of the frame stays 0
    const int lo_x = C::PAD, hi_x = C::PAD + a.g.N, lo_y = C::PAD, hi_y = C::PAD + a.g.M;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int y = y0 + ty * R + r;
      if (y < lo_y || y >= hi_y) continue;
      const ptrdiff_t o = (ptrdiff_t)y * pitch + 3 * xp;
      const float4* fp = reinterpret_cast<const float4*>(a.f + o);
      float fv[12];
#pragma unroll
      for (int j = 0; j < 3; ++j) { const float4 t = fp[j]; fv[4*j] = t.x; fv[4*j+1] = t.y; fv[4*j+2] = t.z; fv[4*j+3] = t.w; }
      float e[12];
#pragma unroll
      for (int f = 0; f < 12; ++f) e[f] = __fsub_rn(acc[r][f], fv[f]);
      if (xp >= lo_x && xp + 3 < hi_x) {
        float4* op = reinterpret_cast<float4*>(a.out + o);
#pragma unroll
        for (int j = 0; j < 3; ++j) op[j] = make_float4(e[4*j], e[4*j+1], e[4*j+2], e[4*j+3]);
      } else {
#pragma unroll
        for (int p = 0; p < 4; ++p)
          if (xp + p >= lo_x && xp + p < hi_x) {
            a.out[o + 3*p] = e[3*p]; a.out[o + 3*p + 1] = e[3*p+1]; a.out[o + 3*p + 2] = e[3*p+2];
          }
      }
    }
  } else {
    // gradu (raw back-projection) over the whole u-frame + reductions for the step size:
    //   g = lambd*gradu + (u-ut)/2.  (pyx:519, float product + exact halving, one rounding)
    uint32_t kg[3] = {0u, 0u, 0u}, ku[3] = {0u, 0u, 0u};
    const float lambd = a.lambd;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int y = y0 + ty * R + r;
      if (y >= a.g.uM) continue;
      const ptrdiff_t o = (ptrdiff_t)y * pitch + 3 * xp;
      const float4* up = reinterpret_cast<const float4*>(a.u + o);
      const float4* tp = reinterpret_cast<const float4*>(a.ut + o);
      float uv[12], tv[12];
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const float4 t = up[j]; uv[4*j] = t.x; uv[4*j+1] = t.y; uv[4*j+2] = t.z; uv[4*j+3] = t.w;
        const float4 s = tp[j]; tv[4*j] = s.x; tv[4*j+1] = s.y; tv[4*j+2] = s.z; tv[4*j+3] = s.w;
      }
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        if (xp + p < a.g.uN) {
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const float g = __fadd_rn(__fmul_rn(lambd, acc[r][3*p+c]), __fmul_rn(__fsub_rn(uv[3*p+c], tv[3*p+c]), 0.5f));
            const uint32_t k1 = key_of(__builtin_fabsf(g));
            const uint32_t k2 = key_of(uv[3*p+c]);
            kg[c] = kg[c] > k1 ? kg[c] : k1;
            ku[c] = ku[c] > k2 ? ku[c] : k2;
          }
        }
      }
      if (xp + 3 < a.g.uN) {
        float4* op = reinterpret_cast<float4*>(a.out + o);
#pragma unroll
        for (int j = 0; j < 3; ++j) op[j] = make_float4(acc[r][4*j], acc[r][4*j+1], acc[r][4*j+2], acc[r][4*j+3]);
      } else {
#pragma unroll
        for (int p = 0; p < 4; ++p)
          if (xp + p < a.g.uN) {
            a.out[o + 3*p] = acc[r][3*p]; a.out[o + 3*p + 1] = acc[r][3*p+1]; a.out[o + 3*p + 2] = acc[r][3*p+2];
          }
      }
    }
    // wave shuffle reduction -> LDS -> one atomic per value per workgroup
#pragma unroll
    for (int c = 0; c < 3; ++c) { kg[c] = wave_max_u32(kg[c]); ku[c] = wave_max_u32(ku[c]); }
    __syncthreads();  // all waves are done reading the tile
    uint32_t* red_lds = reinterpret_cast<uint32_t*>(lds);
    const int wave = tid >> 6, lane = tid & 63;
    if (lane == 0) {
#pragma unroll
      for (int c = 0; c < 3; ++c) { red_lds[wave * 8 + c] = kg[c]; red_lds[wave * 8 + 3 + c] = ku[c]; }
    }
    __syncthreads();
    if (tid < 6) {
      uint32_t m = red_lds[tid];
#pragma unroll
      for (int w = 1; w < 4; ++w) { const uint32_t o = red_lds[w * 8 + tid]; m = m > o ? m : o; }
      const int slot = tid < 3 ? ICS_RED_MAXG + tid : ICS_RED_MAXU + (tid - 3);
      atomicMax(a.red + slot, m);
    }
  }
}

template <int K, int R, int MODE>
hipError_t launch_one(const IcsConvArgs& a, hipStream_t s) {
  using C = ConvCfg<K, R>;
  static bool configured = false;
  auto kern = k_conv<K, R, MODE>;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES);
    if (e != hipSuccess) return e;
    configured = true;
  }
  dim3 grid(a.g.tiles_x, a.g.tiles_y * (ICS_TILE / C::TH));
  hipLaunchKernelGGL(kern, grid, dim3(256), C::LDS_BYTES, s, a);
  return hipGetLastError();
}

template <int K>
hipError_t launch_k(int mode, const IcsConvArgs& a, hipStream_t s) {
  constexpr int R = (K <= 15) ? 4 : 2;
  return mode == 0 ? launch_one<K, R, 0>(a, s) : launch_one<K, R, 1>(a, s);
}

}  // namespace

bool ics_conv_supported(int K) { return K >= 3 && K <= 31 && (K & 1); }

hipError_t ics_launch_conv(int mode, const IcsConvArgs& a, hipStream_t s) {
  switch (a.g.K) {
    case 3: return launch_k<3>(mode, a, s);
    case 5: return launch_k<5>(mode, a, s);
    case 7: return launch_k<7>(mode, a, s);
    case 9: return launch_k<9>(mode, a, s);
    case 11: return launch_k<11>(mode, a, s);
    case 13: return launch_k<13>(mode, a, s);
    case 15: return launch_k<15>(mode, a, s);
    case 17: return launch_k<17>(mode, a, s);
    case 19: return launch_k<19>(mode, a, s);
    case 21: return launch_k<21>(mode, a, s);
    case 23: return launch_k<23>(mode, a, s);
    case 25: return launch_k<25>(mode, a, s);
    case 27: return launch_k<27>(mode, a, s);
    case 29: return launch_k<29>(mode, a, s);
    case 31: return launch_k<31>(mode, a, s);
    default: return hipErrorInvalidValue;
  }
}
