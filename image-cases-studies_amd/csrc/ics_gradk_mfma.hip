// ics_gradk_mfma.hip -- A12+A13 (lib/deconvolution.pyx:567-571), PSF sizes <= 15, on the fp16 matrix cores.
//
//     gradk[a, b, c] = sum_{y,x} E[y, x, c] * U[y + pad - a, x + pad - b, c]            (u-frame coordinates)
//
// Same contraction as k_gradk (ics_kernels.hip): for one residual row y, a 32-pixel chunk of U columns x' and one
// channel,   D[a][b] += sum_k A[a][k] * B[k][b],   A[a][k] = U[y + pad - a][x'0 + k],   B[k][b] = E[y][x'0 + k - pad + b]
// but evaluated with v_mfma_f32_16x16x32_f16 (1.9 PFLOP/s measured, vs 157 TFLOP/s for the fp32 MFMA k_gradk is
// bound by): every fp32 operand is split into two fp16 terms after scaling by a power of two per tile,
//     x * s = hi + lo,   and   A*B ~ Ah*Bh + Ah*Bl + Al*Bh     (22 significand bits per operand, fp32 accumulate),
// exactly as in ics_conv_mfma.hip.  Because the scales change from tile to tile, the MFMA accumulators restart at
// zero for every tile and are folded into fp32 totals with the tile's exact inverse scale.
//
// Tile = 32 residual rows x 64 U columns, all three channels; 4 waves (8 rows each).  LDS (74 KB, two workgroups per
// CU): U rows [y0+pad-15, y0+pad+32) and E rows [y0, y0+32) x columns [x0-8, x0+72) as fp16 hi/lo planes with
// 160-byte rows (conflict-free for the 16 descending lane rows of an A fragment).  The B operand is a Toeplitz
// (sliding) window of an E row: every lane needs 8 consecutive halves starting at half 8g + b + 8 - pad of the chunk's
// 48-half segment; it reads the five dwords that contain them and funnel-shifts by the parity (v_alignbit).
// (A ds_bpermute gather from a row image measured ~5 LDS cycles per bpermute: the kernel was LDS-bound at 63 %.)
// Workgroups are persistent and write one partial block each, reduced in double by k_gradk_reduce (deterministic).
#include "ics_kernels.h"

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));

struct GCfg {
  static constexpr int TH = 32, TW = 64, NT = 16;    // tile: residual rows x U columns; taps per axis (padded)
  static constexpr int NW = 4, NTH = 64 * NW;
  static constexpr int UROWS = TH + NT - 1;          // 47
  static constexpr int ECOLS = TW + 16;              // E columns [x0 - 8, x0 + 72)
  static constexpr int ROWB = 160;                   // bytes per LDS row (both operands)
  static constexpr int UPLANE = UROWS * ROWB, EPLANE = TH * ROWB;
  static constexpr int UOFF = 0, EOFF = 6 * UPLANE;
  static constexpr int DATA = 6 * UPLANE + 6 * EPLANE + 64;   // + slack: lanes of unused taps may over-read a row
  static constexpr size_t LDS_BYTES = DATA + 256;
  static constexpr int UXG = TW / 4, EXG = ECOLS / 4;         // 4-pixel groups per staged row
  static constexpr int UTASK = UROWS * UXG, ETASK = TH * EXG; // 752 + 640
  static constexpr int UIT = (UTASK + NTH - 1) / NTH, EIT = (ETASK + NTH - 1) / NTH;
  static constexpr size_t RED_FLOATS = (size_t)NW * 256;      // cross-wave reduction, one channel at a time
  static_assert(2 * LDS_BYTES <= 160 * 1024, "two workgroups per CU");
};

#define ICS_BUF_WORD3 0x00020000  /* gfx9 raw buffer: DATA_FORMAT = 32 */
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7FFFFFFF, ICS_BUF_WORD3);
}

__device__ __forceinline__ void pow2_scale(float m, float& s, float& inv) {
  const uint32_t e = (__float_as_uint(m) >> 23) & 0xFFu;
  uint32_t sb = 127u;
  if (m > 0.f && e != 255u) { sb = 268u - e; sb = sb > 240u ? 240u : sb; }
  s = __uint_as_float(sb << 23);
  inv = __uint_as_float((254u - sb) << 23);
}

__device__ __forceinline__ float wg_max(float m, float* scr, int wave, int lane) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) m = __builtin_fmaxf(m, __shfl_xor(m, off, 64));
  if (lane == 0) scr[wave] = m;
  __syncthreads();
#pragma unroll
  for (int w = 0; w < GCfg::NW; ++w) m = __builtin_fmaxf(m, scr[w]);
  return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, m)));
}

// 4 pixels (12 floats, HWC) -> hi/lo halves of three planes
__device__ __forceinline__ void split_store(const f32x4u (&v)[3], float s, unsigned char* dst, int plane_bytes) {
  float f[12];
#pragma unroll
  for (int h = 0; h < 3; ++h)
#pragma unroll
    for (int e = 0; e < 4; ++e) f[4 * h + e] = v[h][e] * s;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    h4 hi, lo;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const float x = f[3 * p + c];
      const _Float16 xh = (_Float16)x;
      hi[p] = xh;
      lo[p] = (_Float16)(x - (float)xh);
    }
    *reinterpret_cast<h4*>(dst + (2 * c) * plane_bytes) = hi;
    *reinterpret_cast<h4*>(dst + (2 * c + 1) * plane_bytes) = lo;
  }
}

// requests the staged rows of tile t: U rows [y0 + pad - 15, y0 + pad + 32) x [x0, x0 + 64) and E rows [y0, y0 + 32) x
// [x0 - 8, x0 + 72), one 4-pixel group (three dwordx4) per task
__device__ __forceinline__ void load_tile(f32x4u (&pu)[GCfg::UIT][3], f32x4u (&pe)[GCfg::EIT][3], __amdgpu_buffer_rsrc_t rs_u,
                                          __amdgpu_buffer_rsrc_t rs_e, const IcsGeom& G, int t, int tid) {
  using C = GCfg;
  const int x0 = (t % G.tiles_x) * C::TW, y0 = (t / G.tiles_x) * C::TH, pitch = G.pitch;
  const int su = 4 * ((G.ay + y0 + G.pad - (C::NT - 1)) * pitch + 3 * (G.ax + x0));
#pragma unroll
  for (int k = 0; k < C::UIT; ++k) {
    int v = tid + k * C::NTH; v = v < C::UTASK ? v : C::UTASK - 1;
    const int row = v / C::UXG, xg = v - row * C::UXG;
#pragma unroll
    for (int h = 0; h < 3; ++h)
      pu[k][h] = __builtin_bit_cast(f32x4u, __builtin_amdgcn_raw_buffer_load_b128(rs_u, 4 * (row * pitch + 12 * xg) + 16 * h, su, 0));
  }
  const int se = 4 * ((G.ay + y0) * pitch + 3 * (G.ax + x0 - 8));
#pragma unroll
  for (int k = 0; k < C::EIT; ++k) {
    int v = tid + k * C::NTH; v = v < C::ETASK ? v : C::ETASK - 1;
    const int row = v / C::EXG, xg = v - row * C::EXG;
#pragma unroll
    for (int h = 0; h < 3; ++h)
      pe[k][h] = __builtin_bit_cast(f32x4u, __builtin_amdgcn_raw_buffer_load_b128(rs_e, 4 * (row * pitch + 12 * xg) + 16 * h, se, 0));
  }
}

__device__ __forceinline__ int opaque(int x) { asm volatile("" : "+v"(x)); return x; }

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_gradk_mfma(IcsGradkArgs a) {
  using C = GCfg;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  float* fscr = reinterpret_cast<float*>(lds + C::DATA);
  const IcsGeom& G = a.geo;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lg = lane >> 4;
  const int pad = G.pad, pitch = G.pitch;
  const int ntx = G.tiles_x, nty = G.tiles_y * (ICS_TILE / C::TH);
  const int ntiles = ntx * nty;

  // frames addressed from their allocation start (offsets are then non-negative)
  const ptrdiff_t org = (ptrdiff_t)G.ay * pitch + 3 * G.ax;
  const __amdgpu_buffer_rsrc_t rs_u = make_rsrc(a.u - org), rs_e = make_rsrc(a.e - org);

  // lane constants of the B gather: first half of this lane's slice inside the 48-half segment
  const int jb = li < G.K ? li : G.K - 1;                       // lanes of unused taps repeat the last one
  const int bo = 8 * lg + jb + 8 - pad;
  const uint32_t bsh = (bo & 1) * 16;

  f4 tot[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) tot[c] = (f4){0.f, 0.f, 0.f, 0.f};

  f32x4u pu[C::UIT][3], pe[C::EIT][3];
  if ((int)blockIdx.x < ntiles) load_tile(pu, pe, rs_u, rs_e, G, blockIdx.x, tid);
#pragma unroll 1
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    // ---- the rows of this tile are in registers (requested during the previous tile's MFMA phase) ----------
    float mu = 0.f, me = 0.f;
#pragma unroll
    for (int k = 0; k < C::UIT; ++k)
#pragma unroll
      for (int h = 0; h < 3; ++h)
#pragma unroll
        for (int e = 0; e < 4; ++e) mu = __builtin_fmaxf(mu, __builtin_fabsf(pu[k][h][e]));
#pragma unroll
    for (int k = 0; k < C::EIT; ++k)
#pragma unroll
      for (int h = 0; h < 3; ++h)
#pragma unroll
        for (int e = 0; e < 4; ++e) me = __builtin_fmaxf(me, __builtin_fabsf(pe[k][h][e]));
    __syncthreads();                       // previous tile's planes fully consumed (and fscr free)
    mu = wg_max(mu, fscr, wave, lane);
    me = wg_max(me, fscr + 8, wave, lane);
    float s_u, inv_u, s_e, inv_e;
    pow2_scale(mu, s_u, inv_u);
    pow2_scale(me, s_e, inv_e);
#pragma unroll
    for (int k = 0; k < C::UIT; ++k) {
      const int v = tid + k * C::NTH;
      if (v < C::UTASK) {
        const int row = v / C::UXG, xg = v - row * C::UXG;
        split_store(pu[k], s_u, lds + C::UOFF + row * C::ROWB + 8 * xg, C::UPLANE);
      }
    }
#pragma unroll
    for (int k = 0; k < C::EIT; ++k) {
      const int v = tid + k * C::NTH;
      if (v < C::ETASK) {
        const int row = v / C::EXG, xg = v - row * C::EXG;
        split_store(pe[k], s_e, lds + C::EOFF + row * C::ROWB + 8 * xg, C::EPLANE);
      }
    }
    __syncthreads();

    // next tile's rows: in flight during the whole MFMA phase (which issues no vector-memory load)
    if (t + (int)gridDim.x < ntiles) load_tile(pu, pe, rs_u, rs_e, G, t + (int)gridDim.x, opaque(tid));
    __builtin_amdgcn_sched_barrier(0);

    // ---- MFMA phase: wave w owns residual rows 8w .. 8w+7 ---------------------------------------------------
    // six independent accumulators (channel x chunk): the three split terms of one product never follow each
    // other on the same accumulator
    f4 acc[3][C::TW / 32];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int X = 0; X < C::TW / 32; ++X) acc[c][X] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
    for (int yy = 0; yy < C::TH / C::NW; ++yy) {
      const int y = wave * (C::TH / C::NW) + yy;
      // A: U row (y + 15 - a) of the staged block for lane a, columns 32X + 8g .. +7
      const unsigned char* arow = lds + C::UOFF + (y + C::NT - 1 - li) * C::ROWB + 16 * lg;
      // B: E row y, this lane's 8 halves start at half `bo` of the 48-half segment that starts at column 32X
      const unsigned char* erow = lds + C::EOFF + y * C::ROWB + 4 * (bo >> 1);
      h8 Ah[3][C::TW / 32], Al[3][C::TW / 32], Bh[3][C::TW / 32], Bl[3][C::TW / 32];
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int X = 0; X < C::TW / 32; ++X) {
          Ah[c][X] = *reinterpret_cast<const h8*>(arow + (2 * c) * C::UPLANE + 64 * X);
          Al[c][X] = *reinterpret_cast<const h8*>(arow + (2 * c + 1) * C::UPLANE + 64 * X);
#pragma unroll
          for (int sp = 0; sp < 2; ++sp) {
            // 8 halves from half `bo` of the segment: five dwords from dword bo >> 1, funnel-shifted by the parity
            const uint32_t* ep = reinterpret_cast<const uint32_t*>(erow + (2 * c + sp) * C::EPLANE + 64 * X);
            const uint32_t d0 = ep[0], d1 = ep[1], d2 = ep[2], d3 = ep[3], d4 = ep[4];
            u4 w = {__builtin_amdgcn_alignbit(d1, d0, bsh), __builtin_amdgcn_alignbit(d2, d1, bsh),
                    __builtin_amdgcn_alignbit(d3, d2, bsh), __builtin_amdgcn_alignbit(d4, d3, bsh)};
            (sp ? Bl[c][X] : Bh[c][X]) = __builtin_bit_cast(h8, w);
          }
        }
#pragma unroll
      for (int term = 0; term < 3; ++term)
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
          for (int X = 0; X < C::TW / 32; ++X)
            acc[c][X] = __builtin_amdgcn_mfma_f32_16x16x32_f16(term == 2 ? Al[c][X] : Ah[c][X], term == 1 ? Bl[c][X] : Bh[c][X], acc[c][X], 0, 0, 0);
    }
    const float sc = inv_u * inv_e;   // powers of two
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) tot[c][r] += (acc[c][0][r] + acc[c][1][r]) * sc;
  }

  // ---- cross-wave reduction (fixed order) and partial write, one channel per pass ----------------------------
  float* red = reinterpret_cast<float*>(lds);   // [wave][256]: element (row = 4*lg + r, col = li) at [r*64 + lane]
  float* dst = a.partial + (size_t)blockIdx.x * (3 * C::NT * C::NT);
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave * 256 + r * 64 + lane] = tot[c][r];
    __syncthreads();
    {
      const int v = tid;   // 256 threads, 256 outputs
      float s = red[v];
#pragma unroll
      for (int w = 1; w < C::NW; ++w) s += red[w * 256 + v];   // fixed order -> deterministic
      const int l = v & 63, r = v >> 6;
      const int ta = 4 * (l >> 4) + r, tb = l & 15;
      dst[(c * C::NT + ta) * C::NT + tb] = s;
    }
  }
}

}  // namespace

bool ics_gradk_mfma_supported(int K) { return K >= 3 && K <= 15 && (K & 1); }

hipError_t ics_launch_gradk_mfma(const IcsGradkArgs& a, int nblocks, hipStream_t s) {
  static bool configured[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  if (!configured[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_gradk_mfma), hipFuncAttributeMaxDynamicSharedMemorySize, (int)GCfg::LDS_BYTES);
    if (e != hipSuccess) { (void)hipGetLastError(); return e; }
    configured[dev] = true;
  }
  hipLaunchKernelGGL(k_gradk_mfma, dim3(nblocks), dim3(GCfg::NTH), GCfg::LDS_BYTES, s, a);
  return hipGetLastError();
}
