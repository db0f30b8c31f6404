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
// plus halo is staged once in LDS (dwordx4 global loads, HWC rows are contiguous so a 78-px row
// segment is one 936-B run).  Each lane owns R output rows x 4 pixels (12 floats of the flattened
// x*3+c axis, so the channel of a register is a compile-time constant) and walks the input rows
// once: an LDS row strip is read with ds_read_b128 into registers and feeds all R output rows
// (kernel row a = i - r).  The FMA stream is packed (v_pk_fma_f32, the only way to the fp32 vector
// peak on CDNA4: measured 135 TF/s vs 65 TF/s for v_fmac_f32, csrc/tools/ubench_fma.hip): the two
// halves of an accumulator pair are the SAME pixel/channel of two consecutive output rows, so both
// halves take the same input value (op_sel broadcast of one strip register, no shuffles) and the
// weight operand is the pair (W[a][b][c], W[a-1][b][c]) of two consecutive kernel rows, which is
// wave-uniform and arrives through the scalar cache as an aligned SGPR pair.  The VALU stream is
// therefore pure v_pk_fma_f32 acc, s[w:w+1], v_strip(op_sel), acc.  Each row pair walks K+1 input
// rows (the first/last with one zero weight), a (K+1)/K overhead.  fp32 throughout; the dense
// formulation needed for MFMA would waste >= 50 % of the matrix pipe on the Toeplitz band
// (DESIGN.md), so this path is VALU by design.
#include "ics_common.h"

#ifndef ICS_CONV_NTY32
#define ICS_CONV_NTY32 1
#endif

namespace {

template <int K, int R, int NTY = 16>
struct ConvCfg {
  static constexpr int PAD = K / 2;
  static constexpr int TW = ICS_TILE;
  static constexpr int NT = 16 * NTY;   // threads per workgroup: 16 lanes across x, NTY across y
  static constexpr int TH = NTY * R;
  static constexpr int LROWS = TH + K - 1;
  // LDS row = pixels [x0 - PAD, x0 + TW + PAD): a lane's strip then starts at float 12*tx, 16-B aligned,
  // and is read with ds_read_b128.  (The matching global address is only 4-B aligned: 3*PAD floats.)
  static constexpr int LW_USED = (3 * (TW + 2 * PAD) + 3) & ~3;  // floats staged per LDS row
  static constexpr int LALIGN = 64 / R;                           // R*LWF % 64 == 0: conflict-free b128 strips
  static constexpr int LWF = ((LW_USED + LALIGN - 1) / LALIGN) * LALIGN;
  static constexpr int STRIP = (12 + 3 * (K - 1) + 3) & ~3;
  static constexpr int WROW2 = (6 * K + 3) & ~3;      // floats per packed row-pair weight row (ics_common.h)
  static constexpr size_t LDS_BYTES = (size_t)LROWS * LWF * 4;
  static_assert(12 * 15 + STRIP <= LWF, "strip overruns the LDS row");
  static_assert(ICS_TILE % TH == 0, "tile height must divide the frame granularity");
  static_assert(R % 2 == 0, "output rows are processed in pairs");
};

// 16-byte vector with 4-byte alignment: global_load_dwordx4 only needs dword alignment
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));

typedef float f32x2 __attribute__((ext_vector_type(2)));

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

// LDS row (ap + 2*rp) -> strip registers of row pair rp (ds_read_b128, 16-B aligned by construction)
template <typename C, int R>
__device__ __forceinline__ void load_strips(float (&strip)[R / 2][C::STRIP], const float* lrow0, int ap) {
#pragma unroll
  for (int rp = 0; rp < R / 2; ++rp) {
    const float4* lp = reinterpret_cast<const float4*>(lrow0 + (ap + 2 * rp) * C::LWF);
#pragma unroll
    for (int j = 0; j < C::STRIP / 4; ++j) {
      const float4 t = lp[j];
      strip[rp][4 * j + 0] = t.x; strip[rp][4 * j + 1] = t.y; strip[rp][4 * j + 2] = t.z; strip[rp][4 * j + 3] = t.w;
    }
  }
}

// One packed weight row against the strips of every row pair: (R/2)*K*12 v_pk_fma_f32.
// (every strip element is consumed through op_sel as the low or high half of an aligned VGPR pair)
template <typename C, int K, int R>
__device__ __forceinline__ void fma_row(f32x2 (&A)[R / 2][12], const float (&strip)[R / 2][C::STRIP], const f32x2* __restrict__ wr) {
#pragma unroll
  for (int b = 0; b < K; ++b) {
    const f32x2 w0 = wr[3 * b + 0], w1 = wr[3 * b + 1], w2 = wr[3 * b + 2];
#pragma unroll
    for (int rp = 0; rp < R / 2; ++rp) {
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const float s0 = strip[rp][3 * (p + b) + 0];
        const float s1 = strip[rp][3 * (p + b) + 1];
        const float s2 = strip[rp][3 * (p + b) + 2];
        A[rp][3 * p + 0] = __builtin_elementwise_fma(w0, (f32x2){s0, s0}, A[rp][3 * p + 0]);
        A[rp][3 * p + 1] = __builtin_elementwise_fma(w1, (f32x2){s1, s1}, A[rp][3 * p + 1]);
        A[rp][3 * p + 2] = __builtin_elementwise_fma(w2, (f32x2){s2, s2}, A[rp][3 * p + 2]);
      }
    }
  }
}

template <int K, int R, int MODE, int WPE, int NTY>
__global__ __launch_bounds__(16 * NTY) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_conv(IcsConvArgs a) {
  using C = ConvCfg<K, R, NTY>;
  constexpr int NT = C::NT;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x;
  const int tx = tid & 15, ty = tid >> 4;
  // XCD-aware tile order: workgroup b runs on XCD b % 8 (observed dispatch), each XCD has its own L2.
  // Remap so that every XCD owns one contiguous band of tile rows and the halos shared by neighbouring
  // tiles hit in that XCD's L2 (bijective for any tile count; affects speed only, never results).
  const int gx = a.g.tiles_x, nwg = gridDim.x;
  int lin = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = lin & 7;
    lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (lin >> 3);
  }
  const int tyi = lin / gx, txi = lin - tyi * gx;
  const int x0 = txi * C::TW, y0 = tyi * C::TH;
  const int pitch = a.g.pitch;

  if (MODE != 2) {
  // ---- stage tile + halo: all global loads of a batch are in flight before the LDS writes ----------
  {
    const float* src = a.in + (ptrdiff_t)(y0 - C::PAD) * pitch + 3 * (x0 - C::PAD);
    constexpr int LW4 = C::LW_USED / 4;
    constexpr int NV = C::LROWS * LW4;
    constexpr int NIT = (NV + NT - 1) / NT;
    constexpr int BATCH = NIT;  // one batch: every load of the tile is in flight before the first LDS write
#pragma unroll
    for (int it0 = 0; it0 < NIT; it0 += BATCH) {
      f32x4u val[BATCH];
#pragma unroll
      for (int k = 0; k < BATCH; ++k) {
        if (it0 + k < NIT) {
          int v = tid + (it0 + k) * NT;
          v = v < NV ? v : NV - 1;  // clamp instead of predicating the load
          const int row = v / LW4, c4 = v - row * LW4;
          val[k] = *reinterpret_cast<const f32x4u*>(src + (ptrdiff_t)row * pitch + 4 * c4);
        }
      }
#pragma unroll
      for (int k = 0; k < BATCH; ++k) {
        if (it0 + k < NIT) {
          const int v = tid + (it0 + k) * NT;
          if (v < NV) {
            const int row = v / LW4, c4 = v - row * LW4;
            *reinterpret_cast<float4*>(lds + row * C::LWF + 4 * c4) = make_float4(val[k].x, val[k].y, val[k].z, val[k].w);
          }
        }
      }
    }
  }
  } else {
    // ---- mode 2: the image update of the finished inner iteration (A5/A6/A8/A10, pyx:499-552) is applied
    // while staging: every staged element (tile + halo) is computed from u_old, ut, gradu, image with the
    // reference's float32 rounding, written to LDS, and the tile's own elements are stored to `u_out`.
    // u_old stays intact for the neighbouring tiles' halos (ping-pong), so no grid-wide dependency arises.
    float dt[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float maxu = ics_key2f(a.red[ICS_RED_MAXU + c]);
      const float maxg = ics_key2f(a.red[ICS_RED_MAXG + c]);
      dt[c] = __fdiv_rn(__fmul_rn(a.step, maxu), __fadd_rn(maxg, 1e-15f));
      if (blockIdx.x == 0 && tid == 0) {
        a.scal[ICS_SC_DT + c] = dt[c]; a.scal[ICS_SC_MAXU + c] = maxu; a.scal[ICS_SC_MAXG + c] = maxg;
      }
    }
    uint32_t kmin = 0xFFFFFFFFu, kmax = 0u, knan = 0u;
    const float lambd = a.lambd;
    const ptrdiff_t base = (ptrdiff_t)(y0 - C::PAD) * pitch + 3 * (x0 - C::PAD);
    constexpr int LW4 = C::LW_USED / 4;
    constexpr int NV = C::LROWS * LW4;
    constexpr int NIT = (NV + NT - 1) / NT;
    constexpr int BATCH = 4;
#pragma unroll 1
    for (int it0 = 0; it0 < NIT; it0 += BATCH) {
      f32x4u pu[BATCH], pt[BATCH], pg[BATCH], pf[BATCH];
#pragma unroll
      for (int k = 0; k < BATCH; ++k) {
        int v = tid + (it0 + k) * NT;
        v = v < NV ? v : NV - 1;
        const int row = v / LW4, c4 = v - row * LW4;
        const ptrdiff_t o = base + (ptrdiff_t)row * pitch + 4 * c4;
        pu[k] = *reinterpret_cast<const f32x4u*>(a.in + o);
        pt[k] = *reinterpret_cast<const f32x4u*>(a.ut + o);
        pg[k] = *reinterpret_cast<const f32x4u*>(a.gr + o);
        pf[k] = *reinterpret_cast<const f32x4u*>(a.f + o);
      }
#pragma unroll
      for (int k = 0; k < BATCH; ++k) {
        const int v = tid + (it0 + k) * NT;
        if (v < NV) {
          const int row = v / LW4, c4 = v - row * LW4;
          const int y = y0 - C::PAD + row;
          const bool yin = (y >= C::PAD) && (y < C::PAD + a.g.M);
          const bool yown = (row >= C::PAD) && (row < C::PAD + C::TH) && (y < a.g.uM);
          float un4[4];
          bool own[4];
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            const int fi = 4 * c4 + jj;          // float index inside the LDS row
            const int pxr = fi / 3, c = fi - 3 * pxr;
            const int x = x0 - C::PAD + pxr;
            const float uo = pu[k][jj], to = pt[k][jj], go = pg[k][jj], fo = pf[k][jj];
            const float dtc = c == 0 ? dt[0] : (c == 1 ? dt[1] : dt[2]);
            const float g = __fadd_rn(__fmul_rn(lambd, go), __fmul_rn(__fsub_rn(uo, to), 0.5f));
            float un = __fsub_rn(uo, __fmul_rn(dtc, g));
            own[jj] = yown && (pxr >= C::PAD) && (pxr < C::PAD + C::TW) && (x < a.g.uN);
            if (yin && (x >= C::PAD) && (x < C::PAD + a.g.N)) {
              const float d = ics_dof_ratio(go, fo);
              float D = __fmul_rn(d, d);
              if (!a.blind) D = __fdiv_rn(D, lambd);
              un = __fadd_rn(__fmul_rn(__fsub_rn(1.0f, D), un), __fmul_rn(D, fo));
              if (a.want_dof && own[jj]) {
                if (D != D) knan = 1u;
                else { const uint32_t kk = ics_f2key(D); kmin = kmin < kk ? kmin : kk; kmax = kmax > kk ? kmax : kk; }
              }
            }
            un4[jj] = un;
          }
          *reinterpret_cast<float4*>(lds + row * C::LWF + 4 * c4) = make_float4(un4[0], un4[1], un4[2], un4[3]);
          float* dst = a.u_out + base + (ptrdiff_t)row * pitch + 4 * c4;
          if (own[0] && own[3]) {
            f32x4u w4 = {un4[0], un4[1], un4[2], un4[3]};
            *reinterpret_cast<f32x4u*>(dst) = w4;
          } else {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) if (own[jj]) dst[jj] = un4[jj];
          }
        }
      }
    }
    if (a.want_dof) {
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) {
        const uint32_t o1 = (uint32_t)__shfl_xor((int)kmin, off, 64); kmin = kmin < o1 ? kmin : o1;
        const uint32_t o2 = (uint32_t)__shfl_xor((int)kmax, off, 64); kmax = kmax > o2 ? kmax : o2;
        const uint32_t o3 = (uint32_t)__shfl_xor((int)knan, off, 64); knan = knan > o3 ? knan : o3;
      }
      if ((tid & 63) == 0) {
        if (kmin < a.dofkeys[0]) atomicMin(a.dofkeys + 0, kmin);
        if (kmax > a.dofkeys[1]) atomicMax(a.dofkeys + 1, kmax);
        if (knan) atomicOr(a.dofkeys + 2, 1u);
      }
    }
  }
  __syncthreads();

  // accumulator pairs: A[rp][f] = (output row 2rp, output row 2rp+1) at flat column f
  f32x2 A[R / 2][12];
#pragma unroll
  for (int rp = 0; rp < R / 2; ++rp)
#pragma unroll
    for (int f = 0; f < 12; ++f) A[rp][f] = (f32x2){0.f, 0.f};

  const float* lrow0 = lds + (ty * R) * C::LWF + 12 * tx;
  const f32x2* __restrict__ wbase = reinterpret_cast<const f32x2*>(a.w);

  // Loop over packed weight rows ap = 0..K.  Row ap holds the pairs (W[ap], W[ap-1]) (W[-1] = W[K] = 0)
  // and serves EVERY row pair of the lane: pair rp (output rows 2rp, 2rp+1) takes its input from LDS
  // row ap + 2rp.  The body is branch-free: (R/2) strips, one weight row, (R/2)*K*12 packed FMAs.
  // a wave owns 4*R consecutive output rows; border tiles have waves with no row to produce
  const int wy0 = y0 + __builtin_amdgcn_readfirstlane(tid >> 6) * 4 * R;
  const bool wave_has_rows = (MODE != 1) ? (wy0 < C::PAD + a.g.M && wy0 + 4 * R > C::PAD) : (wy0 < a.g.uM);
  // Software pipeline: K is odd, so the K+1 weight rows are walked in pairs with two strip register sets
  // (ping-pong): the LDS reads of row ap+1 are issued before the FMAs of row ap and land behind them.
  // (Only when both sets fit the VGPR budget of 3 waves/SIMD; otherwise single-buffered.)
  // (budget 176: at 184 / 192 -- K = 23, 25 -- both sets "fit" 256 VGPRs only with 18 .. 59 spills, and a scratch reload in the
  //  weight-row loop costs more than the pipeline buys)
  constexpr bool PINGPONG = (WPE == 2) && ((R / 2) * C::STRIP * 2 + 12 * R <= 176);
  if (PINGPONG) {
    float sa[R / 2][C::STRIP], sb[R / 2][C::STRIP];
    if (wave_has_rows) load_strips<C, R>(sa, lrow0, 0);
#pragma unroll 1
    for (int ap = 0; ap <= (wave_has_rows ? K : -1); ap += 2) {
      load_strips<C, R>(sb, lrow0, ap + 1);
      fma_row<C, K, R>(A, sa, wbase + ap * (C::WROW2 / 2));
      if (ap + 2 <= K) load_strips<C, R>(sa, lrow0, ap + 2);
      fma_row<C, K, R>(A, sb, wbase + (ap + 1) * (C::WROW2 / 2));
    }
  } else {
#pragma unroll 1
    for (int ap = 0; ap <= (wave_has_rows ? K : -1); ++ap) {
      float strip[R / 2][C::STRIP];
      load_strips<C, R>(strip, lrow0, ap);
      fma_row<C, K, R>(A, strip, wbase + ap * (C::WROW2 / 2));
    }
  }
  float acc[R][12];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int f = 0; f < 12; ++f) acc[r][f] = (r & 1) ? A[r / 2][f].y : A[r / 2][f].x;

  // ---- epilogue ----------------------------------------------------------------------------
  const int xp = x0 + 4 * tx;  // first of this lane's 4 pixels (u-frame x)
  if (MODE != 1) {
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
    // float maxima + a NaN flag per lane (2 VALU per value); converted to order-preserving keys once
    float mg[3] = {0.f, 0.f, 0.f}, mu[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    bool nan_g[3] = {false, false, false}, nan_u[3] = {false, false, false}, any = false;
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
            float g;
            if (a.tv_kind >= 2)                                                                  // PAM: G = T + lambd*gradu
              g = (float)((double)a.tv[o + 3*p+c] + (double)__fmul_rn(lambd, acc[r][3*p+c]));
            else if (a.tv_kind == 1 && y >= 1 && y <= a.g.uM - 2 && xp + p >= 1 && xp + p <= a.g.uN - 2)   // active MM-TV, pyx:517
              g = (float)(((double)a.tv[o + 3*p+c] + (double)__fmul_rn(lambd, acc[r][3*p+c])) + (double)__fsub_rn(uv[3*p+c], tv[3*p+c]) / 4.0);
            else
              g = __fadd_rn(__fmul_rn(lambd, acc[r][3*p+c]), __fmul_rn(__fsub_rn(uv[3*p+c], tv[3*p+c]), 0.5f));
            mg[c] = __builtin_fmaxf(mg[c], __builtin_fabsf(g));   // maxnum drops NaN: tracked separately
            mu[c] = __builtin_fmaxf(mu[c], uv[3*p+c]);
            nan_g[c] |= (g != g); nan_u[c] |= (uv[3*p+c] != uv[3*p+c]);
            any = true;
            if (a.tv_kind >= 2) acc[r][3*p+c] = g;   // PAM: the frame written below is G itself (see ics_conv_mfma.hip)
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
    uint32_t kg[3], ku[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      kg[c] = nan_g[c] ? 0xFFC00000u : (any ? ics_f2key(mg[c]) : 0u);   // NaN propagates like np.amax
      ku[c] = nan_u[c] ? 0xFFC00000u : (any ? ics_f2key(mu[c]) : 0u);
      kg[c] = wave_max_u32(kg[c]); ku[c] = wave_max_u32(ku[c]);
    }
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
      for (int w = 1; w < NT / 64; ++w) { const uint32_t o = red_lds[w * 8 + tid]; m = m > o ? m : o; }
      const int slot = tid < 3 ? ICS_RED_MAXG + tid : ICS_RED_MAXU + (tid - 3);
      // the running maximum only grows: a (possibly stale, hence lower) read lets most of the ~8000
      // workgroups skip their atomic instead of serialising on six L2 words
      if (m > a.red[slot]) atomicMax(a.red + slot, m);
    }
  }
}

template <int K, int R, int MODE, int NTY>
hipError_t launch_one(const IcsConvArgs& a, hipStream_t s) {
  using C = ConvCfg<K, R, NTY>;
  static std::atomic<bool> configured[ICS_MAX_DEVICES];  // per device: the dynamic-LDS attribute is a per-device function property
  const int dev = ics_current_device();
  // waves per SIMD the kernel is compiled for: 3 (<= 168 VGPRs, single-buffered strips) when three
  // workgroups fit the LDS, else 2 (<= 256 VGPRs, ping-pong strips)
  constexpr int WPE = (NTY == 32) ? 4 : ((3 * C::LDS_BYTES <= 160 * 1024) ? 3 : 2);
  auto kern = k_conv<K, R, MODE, WPE, NTY>;
  if (hipError_t e = ics_configure_lds(configured, dev, kern, C::LDS_BYTES); e != hipSuccess) return e;
  dim3 grid(a.g.tiles_x * a.g.tiles_y * (ICS_TILE / C::TH));
  hipLaunchKernelGGL(kern, grid, dim3(C::NT), C::LDS_BYTES, s, a);
  return hipGetLastError();
}

template <int K>
hipError_t launch_k(int mode, const IcsConvArgs& a, hipStream_t s) {
  // NTY = 32 (512 threads, 64x64-px tile, 2 workgroups = 16 waves per CU) when its LDS image fits twice
  constexpr int NTY = (2 * ConvCfg<K, 2, 32>::LDS_BYTES <= 160 * 1024 && ICS_CONV_NTY32) ? 32 : 16;
  if (mode == 2) {
    if constexpr (K <= 31) return launch_one<K, 2, 2, NTY>(a, s);
    else return hipErrorInvalidValue;  // the opt-in fused kernel is only built for PSF sizes <= 31
  }
  // R = 2 (64x32-px tiles): 44 KB of LDS at K = 15 -> 3 workgroups per CU; measured 3-6 % faster than
  // R = 4 (64x64 tiles, 2 workgroups per CU) at 4096^2 despite the larger halo (profiles/)
  constexpr int R = 2;
  return mode == 0 ? launch_one<K, R, 0, NTY>(a, s) : launch_one<K, R, 1, NTY>(a, s);
}

}  // namespace

bool ics_conv_supported(int K) { return K >= 3 && K <= 63 && (K & 1); }

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
    case 33: return launch_k<33>(mode, a, s);
    case 35: return launch_k<35>(mode, a, s);
    case 37: return launch_k<37>(mode, a, s);
    case 39: return launch_k<39>(mode, a, s);
    case 41: return launch_k<41>(mode, a, s);
    case 43: return launch_k<43>(mode, a, s);
    case 45: return launch_k<45>(mode, a, s);
    case 47: return launch_k<47>(mode, a, s);
    case 49: return launch_k<49>(mode, a, s);
    case 51: return launch_k<51>(mode, a, s);
    case 53: return launch_k<53>(mode, a, s);
    case 55: return launch_k<55>(mode, a, s);
    case 57: return launch_k<57>(mode, a, s);
    case 59: return launch_k<59>(mode, a, s);
    case 61: return launch_k<61>(mode, a, s);
    case 63: return launch_k<63>(mode, a, s);
    default: return hipErrorInvalidValue;
  }
}
