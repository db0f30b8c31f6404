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
//
// Tile walk (round 2): a workgroup owns a contiguous run of tiles in COLUMN-MAJOR order, i.e. it walks down a 64-column strip.
// Consecutive tiles of a strip share NT - 1 of their UROWS staged rows of u: those rows stay in LDS (moved to the top of the
// planes and rescaled by the ratio of the two tiles' power-of-two scales, which is exact in fp16 barring underflow of the lo
// terms), and only TH new rows are requested, converted and stored.  A 32-row tile staged 47 rows of u per 32 residual rows
// (16-row tiles of the 2 x 2-block kernel: 47 per 16): the PSF gradient moved 1.63x its algorithmic bytes (round-1 verdict);
// with the strip walk u is read ~1.03x.  The scale of a tile covers the new rows and the maximum of the carried rows (tracked
// per tile over exactly those rows, so a bright pixel does not dictate the scale of the tiles below it).
#include "ics_kernels.h"
#include <type_traits>

#ifndef ICS_GRADK_U_AUX
#define ICS_GRADK_U_AUX 0   /* cache policy of the tile loads (2 = nt) */
#endif
#ifndef ICS_GRADK_E_AUX
#define ICS_GRADK_E_AUX 0
#endif

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));

#ifndef ICS_GRADK_SLICE
#define ICS_GRADK_SLICE 12   /* priority slices between the two workgroups of a CU, 2^12 x 10 ns (ics_prio_turn, ics_common.h); 0: off.  6144^2 / 31x31: 0.782 (off) -> 0.755 (2^10) -> 0.747 ms (2^12); 4096^2 / 23x23 0.359 -> 0.342 */
#endif
template <int NB>
struct GCfg {
  static constexpr int TW = 64, NT = 16 * NB;        // U columns per tile; taps per axis (padded to 16 NB)
  // (Round 4 also built 32-row tiles with eight waves and one workgroup per CU for NB = 2 -- half the barriers, staging and carried-row
  //  moves per MFMA: 0.802 vs 0.807 ms at 6144^2 / 31 x 31 on the same box, i.e. nothing; the per-tile costs are not what bounds the
  //  kernel, the 2.5 LDS / funnel-shift instructions per MFMA of its sliding B windows are.  Not kept.)
  static constexpr int TH = NB == 1 ? 32 : 16;       // residual rows per tile
  static constexpr int NW = 4, NTH = 64 * NW;
  static constexpr int WGS = 2;                      // workgroups per CU
  static constexpr int UROWS = TH + NT - 1;          // 47
  static constexpr int ECOLS = TW + 16 * NB;         // E columns [x0 - 8 NB, x0 + 64 + 8 NB)
  static constexpr int UROWB = 160;                  // bytes per LDS row of U: conflict-free for the 16 descending lane rows
  static constexpr int EROWB = 2 * ECOLS;
  static constexpr int UPLANE = UROWS * UROWB, EPLANE = TH * EROWB;
  static constexpr int UOFF = 0, EOFF = 6 * UPLANE;
  // (the E planes of a channel are interleaved dword by dword -- hi dword d at 2d, lo dword d at 2d + 1 -- so that one
  //  8-byte LDS read fetches both split terms of the sliding window: 10 ds_read_b64 instead of 20 ds_read_b32 per
  //  row and channel; the kernel is bound by LDS-array cycles)
  static constexpr int DATA = 6 * UPLANE + 6 * EPLANE + 128;   // + slack: the five-dword B read may overshoot a row
  static constexpr size_t RED_BYTES = (size_t)NW * NB * NB * 256 * 4;   // cross-wave reduction, one channel at a time
  static constexpr size_t LDS_BYTES = (DATA > (int)RED_BYTES ? DATA : (int)RED_BYTES) + 256;
  static constexpr int SCR = (int)LDS_BYTES - 256;
  static constexpr int UXG = TW / 4, EXG = ECOLS / 4;         // 4-pixel groups per staged row
  static constexpr int UTASK = UROWS * UXG, ETASK = TH * EXG;
  static constexpr int UIT = (UTASK + NTH - 1) / NTH, EIT = (ETASK + NTH - 1) / NTH;
  static_assert(WGS * LDS_BYTES <= 160 * 1024, "workgroups per CU");
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
  m = ics_wave_max_f32(m);
  if (lane == 0) scr[wave] = m;
  __syncthreads();
#pragma unroll
  for (int w = 0; w < 4; ++w) m = __builtin_fmaxf(m, scr[w]);
  return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, m)));
}

// three workgroup maxima behind ONE barrier (u scale, residual scale, carried-row bound)
template <int NW>
__device__ __forceinline__ void wg_max3(float& a, float& b, float& c, float* scr, int wave, int lane) {
  a = ics_wave_max_f32(a); b = ics_wave_max_f32(b); c = ics_wave_max_f32(c);
  if (lane == 0) { scr[wave] = a; scr[8 + wave] = b; scr[16 + wave] = c; }
  __syncthreads();
#pragma unroll
  for (int w = 0; w < NW; ++w) { a = __builtin_fmaxf(a, scr[w]); b = __builtin_fmaxf(b, scr[8 + w]); c = __builtin_fmaxf(c, scr[16 + w]); }
  a = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, a)));
  b = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, b)));
  c = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, c)));
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

// 4 pixels (12 floats, HWC) -> one 16-byte group {hi d0, lo d0, hi d1, lo d1} per channel of the interleaved E planes
__device__ __forceinline__ void split_store_interleaved(const f32x4u (&v)[3], float s, unsigned char* dst, int chan_bytes) {
  float f[12];
#pragma unroll
  for (int h = 0; h < 3; ++h)
#pragma unroll
    for (int e = 0; e < 4; ++e) f[4 * h + e] = v[h][e] * s;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    _Float16 hi[4], lo[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const float x = f[3 * p + c];
      hi[p] = (_Float16)x;
      lo[p] = (_Float16)(x - (float)hi[p]);
    }
    const h8 w = {hi[0], hi[1], lo[0], lo[1], hi[2], hi[3], lo[2], lo[3]};
    *reinterpret_cast<h8*>(dst + c * chan_bytes) = w;
  }
}

// requests the staged rows of tile t: U rows [y0 + pad - NT + 1, y0 + pad + TH) x [x0, x0 + 64) and E rows [y0, y0 + TH) x
// [x0 - 8 NB, x0 + 64 + 8 NB), one 4-pixel group (three dwordx4) per task
// `t` is a column-major tile index (strip = t / nty, row block = t % nty).  carry: the first NT - 1 rows of the U block are
// already in LDS (from the tile above): only rows NT - 1 .. UROWS - 1 are requested, as tasks 0 .. TH * UXG - 1.
// PL (round 5): both frames are channel-planar mirrors (ics_common.h; the FFT-tile pipeline keeps u and e that way): a 4-pixel group is one
// dwordx4 per plane, rearranged into the HWC order the split expects (register renaming)
template <bool PL, int aux>
__device__ __forceinline__ void load_group(f32x4u (&v)[3], __amdgpu_buffer_rsrc_t rs, int voff, int soff, int plane_bytes) {
  if (!PL) {
#pragma unroll
    for (int h = 0; h < 3; ++h) v[h] = __builtin_bit_cast(f32x4u, __builtin_amdgcn_raw_buffer_load_b128(rs, voff + 16 * h, soff, aux));
  } else {
    f32x4u p[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) p[c] = __builtin_bit_cast(f32x4u, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff + c * plane_bytes, aux));
#pragma unroll
    for (int i = 0; i < 12; ++i) v[i >> 2][i & 3] = p[i % 3][i / 3];   // HWC float i = pixel i / 3, channel i % 3
  }
}

template <int NB, bool PL>
__device__ __forceinline__ void load_tile(f32x4u (&pu)[GCfg<NB>::UIT][3], f32x4u (&pe)[GCfg<NB>::EIT][3], __amdgpu_buffer_rsrc_t rs_u,
                                          __amdgpu_buffer_rsrc_t rs_e, const IcsGeom& G, int nty, int t, bool carry, int tid) {
  using C = GCfg<NB>;
  const int x0 = (t / nty) * C::TW, y0 = (t % nty) * C::TH, pitch = PL ? ics_ppitch(G) : G.pitch;
  constexpr int XM = PL ? 1 : 3;                       // floats per pixel step
  const int plane_bytes = PL ? 4 * G.rows * pitch : 0;
  const int r0 = carry ? C::NT - 1 : 0;
  const int ntask = (C::UROWS - r0) * C::UXG;
  const int su = 4 * ((G.ay + y0 + G.pad - (C::NT - 1) + r0) * pitch + XM * (G.ax + x0));
#pragma unroll
  for (int k = 0; k < C::UIT; ++k) {
    if (k * C::NTH >= ntask) break;                       // wave-uniform
    int v = tid + k * C::NTH; v = v < ntask ? v : ntask - 1;
    const int row = v / C::UXG, xg = v - row * C::UXG;
    load_group<PL, ICS_GRADK_U_AUX>(pu[k], rs_u, 4 * (row * pitch + 4 * XM * xg), su, plane_bytes);
  }
  const int se = 4 * ((G.ay + y0) * pitch + XM * (G.ax + x0 - 8 * NB));
#pragma unroll
  for (int k = 0; k < C::EIT; ++k) {
    int v = tid + k * C::NTH; v = v < C::ETASK ? v : C::ETASK - 1;
    const int row = v / C::EXG, xg = v - row * C::EXG;
    load_group<PL, ICS_GRADK_E_AUX>(pe[k], rs_e, 4 * (row * pitch + 4 * XM * xg), se, plane_bytes);
  }
}

__device__ __forceinline__ int opaque(int x) { asm volatile("" : "+v"(x)); return x; }

template <int NB, bool PL>
__global__ __launch_bounds__(GCfg<NB>::NTH) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_gradk_mfma(IcsGradkArgs a) {
  using C = GCfg<NB>;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  float* fscr = reinterpret_cast<float*>(lds + C::SCR);
  const IcsGeom& G = a.geo;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lg = lane >> 4;
  const int pad = G.pad, pitch = G.pitch;
  const int ntx = G.tiles_x, nty = G.tiles_y * (ICS_TILE / C::TH);
  const int ntiles = ntx * nty;

  // frames addressed from their allocation start (offsets are then non-negative)
  const ptrdiff_t org = PL ? (ptrdiff_t)G.ay * ics_ppitch(G) + G.ax : (ptrdiff_t)G.ay * pitch + 3 * G.ax;
  const __amdgpu_buffer_rsrc_t rs_u = make_rsrc(a.u - org), rs_e = make_rsrc(a.e - org);

  // lane constants of the B operand: for tap block jb this lane's 8 halves start at half bo of the row segment that
  // begins at column 32X (plane column 0 = frame column x0 - 8 NB); lanes of unused taps repeat the last one
  uint32_t boff[NB], bsh[NB];
#pragma unroll
  for (int jb = 0; jb < NB; ++jb) {
    const int tap = 16 * jb + li < G.K ? 16 * jb + li : G.K - 1;
    const int bo = 8 * lg + tap + 8 * NB - pad;
    boff[jb] = 8u * (uint32_t)(bo >> 1);   // interleaved planes: 8 bytes per dword index
    bsh[jb] = (uint32_t)(bo & 1) * 16u;
  }

  f4 tot[3][NB][NB];
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int ia = 0; ia < NB; ++ia)
#pragma unroll
      for (int jb = 0; jb < NB; ++jb) tot[c][ia][jb] = (f4){0.f, 0.f, 0.f, 0.f};

  const int team = (int)gridDim.x >= C::WGS ? (int)blockIdx.x / ((int)gridDim.x / C::WGS) % C::WGS : 0;   // which of the CU's workgroups (ics_prio_turn)
  // this workgroup's run of tiles, column-major: [t0, t1)
  const int t0 = (int)((long)ntiles * blockIdx.x / gridDim.x), t1 = (int)((long)ntiles * (blockIdx.x + 1) / gridDim.x);
  f32x4u pu[C::UIT][3], pe[C::EIT][3];
  if (t0 < t1) load_tile<NB, PL>(pu, pe, rs_u, rs_e, G, nty, t0, false, tid);
  float mu_prev = 0.f, mc_last = 0.f, s_prev = 1.f;
#pragma unroll 1
  for (int t = t0; t < t1; ++t) {
    // the tile above in the same strip was the previous tile of this workgroup: its last NT - 1 rows of u are this tile's first
    const bool carry = t > t0 && (t % nty) != 0;
    const int r0 = carry ? C::NT - 1 : 0;
    const int ntask = (C::UROWS - r0) * C::UXG;
    // ---- the rows of this tile are in registers (requested during the previous tile's MFMA phase) ----------
    // mc: maximum over the rows in registers that the NEXT tile of the strip inherits (staged rows >= TH).  It is the carried bound of the next tile's scale: with the maximum of the whole
    // tile instead, one bright pixel at the top of a strip set the split scale of every tile below it (round-2 advice).
    float mu = carry ? mu_prev : 0.f, me = 0.f, mc = 0.f;
#pragma unroll
    for (int k = 0; k < C::UIT; ++k)
      if (k * C::NTH < ntask) {
        float mk = 0.f;
#pragma unroll
        for (int h = 0; h < 3; ++h)
#pragma unroll
          for (int e = 0; e < 4; ++e) mk = __builtin_fmaxf(mk, __builtin_fabsf(pu[k][h][e]));
        mu = __builtin_fmaxf(mu, mk);
        const int v = tid + k * C::NTH;
        if (v < ntask && r0 + v / C::UXG >= C::TH) mc = __builtin_fmaxf(mc, mk);
      }
#pragma unroll
    for (int k = 0; k < C::EIT; ++k)
#pragma unroll
      for (int h = 0; h < 3; ++h)
#pragma unroll
        for (int e = 0; e < 4; ++e) me = __builtin_fmaxf(me, __builtin_fabsf(pe[k][h][e]));
    __syncthreads();                       // previous tile's planes fully consumed (and fscr free)
    ics_prio_turn(ICS_GRADK_SLICE, team, C::WGS);
    wg_max3<C::NW>(mu, me, mc, fscr, wave, lane);
    float s_u, inv_u, s_e, inv_e;
    pow2_scale(mu, s_u, inv_u);
    pow2_scale(me, s_e, inv_e);
    if (carry) {
      // rows [TH, TH + NT - 1) of the six planes -> rows [0, NT - 1), times s_u / s_prev (a power of two: exact in fp16 unless a
      // lo term drops below the subnormal quantum).  In passes of TH rows: the destination rows of a pass are source rows of
      // EARLIER passes only (dst [p TH, (p+1) TH) <- src [(p+1) TH, (p+2) TH)), so a barrier between the passes is enough.
      constexpr int CH16 = C::UROWB / 16;                         // 16-byte pieces per row
      // (the ratio itself need not fit fp16: a strip whose first tiles are all zero -- scale 1 -- and whose next tile holds values
      //  below 0.5 -- scale 2^16 -- gave ratio = inf and inf * 0 = NaN in every carried zero.  Outside fp16's power-of-two range
      //  the rows are rescaled through fp32; the bound on the carried rows keeps every product finite.)
      const float ratio_f = s_u / s_prev;
      const bool wide = !(ratio_f <= 32768.f && ratio_f >= 6.103515625e-05f);
      const _Float16 ratio = wide ? (_Float16)1.f : (_Float16)ratio_f;
#pragma unroll
      for (int p0 = 0; p0 < C::NT - 1; p0 += C::TH) {
        const int nrow = (C::NT - 1 - p0) < C::TH ? (C::NT - 1 - p0) : C::TH;
        const int npiece = 6 * nrow * CH16;
        if (p0 > 0) __syncthreads();
        for (int v = tid; v < npiece; v += C::NTH) {
          const int pl = v / (nrow * CH16), rem = v - pl * nrow * CH16;
          unsigned char* base = lds + C::UOFF + pl * C::UPLANE + p0 * C::UROWB + rem * 16;
          h8 val = *reinterpret_cast<const h8*>(base + C::TH * C::UROWB);
          if (wide) {   // workgroup-uniform, rare
#pragma unroll
            for (int i = 0; i < 8; ++i) val[i] = (_Float16)((float)val[i] * ratio_f);
          } else {
            val = val * ratio;
          }
          *reinterpret_cast<h8*>(base) = val;
        }
      }
    }
#pragma unroll
    for (int k = 0; k < C::EIT; ++k) {
      const int v = tid + k * C::NTH;
      if (v < C::ETASK) {
        const int row = v / C::EXG, xg = v - row * C::EXG;
        split_store_interleaved(pe[k], s_e, lds + C::EOFF + row * (2 * C::EROWB) + 16 * xg, 2 * C::EPLANE);
      }
    }
    // The new rows land on [NT - 1, UROWS), which contains the SOURCE rows [TH + p0, TH + NT - 1) of the last carry pass: every
    // wave must have finished reading them (round-2 advice: without this barrier a wave that was done early overwrote rows a
    // lagging wave had not carried yet -- a mix of old and new u in the gradient, rarely and silently).  The residual planes
    // above are independent of the carry and are written in front of the barrier.
    if (carry) __syncthreads();
#pragma unroll
    for (int k = 0; k < C::UIT; ++k) {
      const int v = tid + k * C::NTH;
      if (v < ntask) {
        const int row = r0 + v / C::UXG, xg = v % C::UXG;
        split_store(pu[k], s_u, lds + C::UOFF + row * C::UROWB + 8 * xg, C::UPLANE);
      }
    }
    // (were TH < NT - 1, a row would be carried through two tiles: the inherited rows would stem from the new rows of this tile and
    //  of the one before -- the 16-row tiles of the 2 x 2-block kernel until round 3)
    mu_prev = C::TH >= C::NT - 1 ? mc : __builtin_fmaxf(mc, carry ? mc_last : 0.f);
    mc_last = mc; s_prev = s_u;
    __syncthreads();

    // next tile's rows: in flight during the whole MFMA phase (which issues no vector-memory load)
    if (t + 1 < t1) load_tile<NB, PL>(pu, pe, rs_u, rs_e, G, nty, t + 1, ((t + 1) % nty) != 0, opaque(tid));
    __builtin_amdgcn_sched_barrier(0);

    // ---- MFMA phase: wave w owns TH/4 consecutive residual rows ------------------------------------------------
    // independent accumulators (channel x chunk x tap blocks): the three split terms of one product never follow
    // each other on the same accumulator
    constexpr int NX = C::TW / 32;
    constexpr int AX = NB == 1 ? NX : 1;   // accumulator sets along the chunk axis (registers: NB = 2 has 12 blocks already)
    f4 acc[3][AX][NB][NB];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int X = 0; X < AX; ++X)
#pragma unroll
        for (int ia = 0; ia < NB; ++ia)
#pragma unroll
          for (int jb = 0; jb < NB; ++jb) acc[c][X][ia][jb] = (f4){0.f, 0.f, 0.f, 0.f};
    // Software pipeline over the steps.  NB = 1: a step is (residual row, channel) with both 32-column chunks, two steps deep -- the
    // LDS operands of step i + 2 are requested, and the funnel shifts of step i + 1 done, in the shadows of the MFMAs of step i.
    // NB = 2 (round 4): 12 accumulator blocks leave no registers for that, and until round 3 its operands were requested and used
    // in the same step -- the wave waited out the LDS latency in front of every group of MFMAs (matrix pipe 41 % busy at 6144^2,
    // 31 x 31: profiles/r04_6144_31_before_sq2.txt).  A step is now (residual row, channel, ONE chunk): the operands of step i + 1
    // are requested in front of the 12 MFMAs of step i and shifted behind them; with single-chunk steps that costs no register
    // more than before (A fragments double-buffered, raw B dwords single, finished B fragments double).
    // The reads are volatile: plain loads were sunk to their first use.
    typedef uint32_t u2 __attribute__((ext_vector_type(2)));
    constexpr int XS = NB == 1 ? NX : 1;                   // chunks per step
    constexpr int NSTEP = 3 * (C::TH / C::NW) * (NX / XS);
    constexpr int DEPTH = NB == 1 ? 2 : 1;                 // steps between request and use
    typedef const volatile __attribute__((address_space(3))) u2* lds_u2p;
    typedef const volatile __attribute__((address_space(3))) u4* lds_u4p;
    constexpr int RA = DEPTH + 1, RB = DEPTH, BD = 2;      // buffers of A fragments / of raw B dwords / of finished B fragments
    u4 rAh[RA][XS][NB], rAl[RA][XS][NB];
    u2 rB[RB][XS][NB][5];
    h8 Bh[BD][XS][NB], Bl[BD][XS][NB];
    auto issue = [&](int i) {
      const int rc = i / (NX / XS), X0 = (i % (NX / XS)) * XS;      // (row, channel) index and first chunk of the step
      const int y = wave * (C::TH / C::NW) + rc / 3, c = rc % 3, pa = i % RA, pb = i % RB;
      // A: U row (y + NT - 1 - a) of the staged block for tap a = 16 ia + lane row, columns 32X + 8g .. +7
      const uint32_t arow = (uint32_t)(uintptr_t)(lds_u4p)(lds + C::UOFF) + (uint32_t)((y + C::NT - 1 - li) * C::UROWB + 16 * lg);
      // B: E row y, this lane's 8 halves start at half `bo` of the segment that starts at column 32X
      const uint32_t erow = (uint32_t)(uintptr_t)(lds_u2p)(lds + C::EOFF) + (uint32_t)(y * (2 * C::EROWB) + c * (2 * C::EPLANE));
#pragma unroll
      for (int X = 0; X < XS; ++X) {
#pragma unroll
        for (int ia = 0; ia < NB; ++ia) {
          rAh[pa][X][ia] = *reinterpret_cast<lds_u4p>(arow - (uint32_t)(16 * ia * C::UROWB) + (uint32_t)((2 * c) * C::UPLANE + 64 * (X0 + X)));
          rAl[pa][X][ia] = *reinterpret_cast<lds_u4p>(arow - (uint32_t)(16 * ia * C::UROWB) + (uint32_t)((2 * c + 1) * C::UPLANE + 64 * (X0 + X)));
        }
#pragma unroll
        for (int jb = 0; jb < NB; ++jb) {
          const lds_u2p ep = reinterpret_cast<lds_u2p>(erow + boff[jb] + (uint32_t)(128 * (X0 + X)));
#pragma unroll
          for (int d = 0; d < 5; ++d) rB[pb][X][jb][d] = ep[d];
        }
      }
    };
    // 8 halves from half `bo` of the segment: five (hi, lo) dword pairs from dword bo >> 1, funnel-shifted by the parity
    auto finish = [&](int i) {
#pragma unroll
      for (int X = 0; X < XS; ++X)
#pragma unroll
        for (int jb = 0; jb < NB; ++jb) {
          const u2* d = rB[i % RB][X][jb];
          const u4 wh = {__builtin_amdgcn_alignbit(d[1].x, d[0].x, bsh[jb]), __builtin_amdgcn_alignbit(d[2].x, d[1].x, bsh[jb]),
                         __builtin_amdgcn_alignbit(d[3].x, d[2].x, bsh[jb]), __builtin_amdgcn_alignbit(d[4].x, d[3].x, bsh[jb])};
          const u4 wl = {__builtin_amdgcn_alignbit(d[1].y, d[0].y, bsh[jb]), __builtin_amdgcn_alignbit(d[2].y, d[1].y, bsh[jb]),
                         __builtin_amdgcn_alignbit(d[3].y, d[2].y, bsh[jb]), __builtin_amdgcn_alignbit(d[4].y, d[3].y, bsh[jb])};
          Bh[i % BD][X][jb] = __builtin_bit_cast(h8, wh);
          Bl[i % BD][X][jb] = __builtin_bit_cast(h8, wl);
        }
    };
    if (DEPTH == 2) { issue(0); issue(1); finish(0); } else { issue(0); finish(0); }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NSTEP; ++i) {
      const int c = (i / (NX / XS)) % 3, X0 = (i % (NX / XS)) * XS;
      if (DEPTH == 2 && i + 1 < NSTEP) finish(i + 1);   // requested a step ago
      if (i + DEPTH < NSTEP) issue(i + DEPTH);
#pragma unroll
      for (int term = 0; term < 3; ++term)
#pragma unroll
        for (int X = 0; X < XS; ++X)
#pragma unroll
          for (int ia = 0; ia < NB; ++ia)
#pragma unroll
            for (int jb = 0; jb < NB; ++jb) {
              const h8 av = __builtin_bit_cast(h8, term == 2 ? rAl[i % RA][X][ia] : rAh[i % RA][X][ia]);
              acc[c][(X0 + X) % AX][ia][jb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, term == 1 ? Bl[i % BD][X][jb] : Bh[i % BD][X][jb], acc[c][(X0 + X) % AX][ia][jb], 0, 0, 0);
            }
      if (DEPTH == 1 && i + 1 < NSTEP) finish(i + 1);   // requested in front of this step's MFMAs
      if (DEPTH == 2) {
        // issue order of a step: per MFMA (6) two or three of the 14 reads of step i + 2 and three of the 16 shifts of step i + 1
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          if (i + 2 < NSTEP) {
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            if (k < 2) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          }
          if (i + 1 < NSTEP) {
            __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
            if (k < 4) __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
          }
        }
      } else {
        // 12 MFMAs: the 14 reads of step i + 1 behind the first seven (two each), its 16 shifts behind the last four (four each)
#pragma unroll
        for (int k = 0; k < 12; ++k) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          if (i + 1 < NSTEP) {
            if (k < 7) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            if (k >= 8) __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    const float sc = inv_u * inv_e;   // powers of two
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int ia = 0; ia < NB; ++ia)
#pragma unroll
        for (int jb = 0; jb < NB; ++jb)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float v = 0.f;
#pragma unroll
            for (int X = 0; X < AX; ++X) v += acc[c][X][ia][jb][r];
            tot[c][ia][jb][r] += v * sc;
          }
  }

  // ---- cross-wave reduction (fixed order) and partial write, one channel per pass ----------------------------
  float* red = reinterpret_cast<float*>(lds);   // [wave][ia][jb][256]: element (row = 4*lg + r, col = li) at [r*64 + lane]
  constexpr int PER = NB * NB * 256;
  float* dst = a.partial + (size_t)blockIdx.x * (3 * C::NT * C::NT);
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    __syncthreads();
#pragma unroll
    for (int ia = 0; ia < NB; ++ia)
#pragma unroll
      for (int jb = 0; jb < NB; ++jb)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[((wave * NB + ia) * NB + jb) * 256 + r * 64 + lane] = tot[c][ia][jb][r];
    __syncthreads();
    for (int v = tid; v < PER; v += C::NTH) {
      float s = red[v];
#pragma unroll
      for (int w = 1; w < C::NW; ++w) s += red[w * PER + v];   // fixed order -> deterministic
      const int l = v & 63, r = (v >> 6) & 3, blk = v >> 8;
      const int jb = blk % NB, ia = blk / NB;
      const int ta = 16 * ia + 4 * (l >> 4) + r, tb = 16 * jb + (l & 15);
      dst[(c * C::NT + ta) * C::NT + tb] = s;
    }
  }
}

template <int NB, bool PL>
hipError_t launch_nb(const IcsGradkArgs& a, int nblocks, hipStream_t s) {
  using C = GCfg<NB>;
  static std::atomic<bool> configured[ICS_MAX_DEVICES];
  const int dev = ics_current_device();
  auto kern = k_gradk_mfma<NB, PL>;
  if (hipError_t e = ics_configure_lds(configured, dev, kern, C::LDS_BYTES); e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3(nblocks), dim3(C::NTH), C::LDS_BYTES, s, a);
  return hipGetLastError();
}

}  // namespace

bool ics_gradk_mfma_supported(int K) { return K >= 3 && K <= 31 && (K & 1); }

hipError_t ics_launch_gradk_mfma(const IcsGradkArgs& a, int nblocks, hipStream_t s) {
  if (a.planar) return a.geo.K <= 15 ? launch_nb<1, true>(a, nblocks, s) : launch_nb<2, true>(a, nblocks, s);
  return a.geo.K <= 15 ? launch_nb<1, false>(a, nblocks, s) : launch_nb<2, false>(a, nblocks, s);
}
