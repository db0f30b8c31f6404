// ics_synth_gradk_mfma.hip -- A11 + A12 + A13 of the blind inner iteration in ONE pass over u (PSF sizes 3..15):
//
//   e'    = convolve(u, psf, "valid") - image                      lib/deconvolution.pyx:555-565   (A11)
//   gradk = convolve(rot180(u), e', "valid")                       lib/deconvolution.pyx:567-571   (A12 + A13)
//         = sum_{y,x} e'[y, x, c] * u[y + pad - a, x + pad - b, c]                (u-frame coordinates)
//
// Why fuse: as two kernels the residual e' makes a round trip through HBM (written by k_conv_mfma<K,0>, re-read with a
// 16-column halo by k_gradk_mfma) and u is staged and split into fp16 planes twice; the PSF gradient alone moved 1.63x its
// algorithmic bytes (profiles/r01_hbm_traffic.json).  The u planes a convolution tile holds in LDS are exactly the A operand
// of the gradient restricted to that tile's residual pixels, so the tile sums need nothing from neighbouring tiles.  Here e'
// never leaves the CU (it is written to HBM only where the stop-test statistics read it, A18/A19 pyx:593-638), u is read
// once: 2 frame transits (u, image) instead of 5.3.
//
// Shape.  Persistent 4-wave workgroups, two per CU, 64x64-pixel tiles of the M x N interior, same walk, same fp16-split
// arithmetic and the same Toeplitz convolution loop as ics_conv_mfma.hip (see there), but ONE CHANNEL AT A TIME through LDS so
// that two workgroups still fit a CU:
//   * the fp32 HWC rows of the tile stay in registers (84 VGPRs) until the third channel is converted; the next tile's rows
//     are requested right after that and are in flight during the last two matrix phases;
//   * LDS (77 KB at K = 15): two buffers of (hi, lo) u planes for one channel (rows grouped by y mod 4 as in ics_conv_mfma.hip), the e'
//     planes of one channel (64 rows x 80 halves, hi/lo interleaved dword by dword, zero columns left and right so that the
//     sliding windows of the tile's border need no neighbour), the convolution weights;
//   * per channel c:  conv(c) -> e'(c) in registers -> [barrier: tile maximum of |e'|] -> e'(c) planes, u planes of c+1 ->
//     [barrier] -> gradk(c), conv(c+1) ...: two barriers per channel, and a wave runs gradk(c) and conv(c+1) back to back so
//     the four waves drift apart and overlap their LDS-heavy (gradient) and matrix-heavy (convolution) phases;
//   * gradient step = one residual row y, one channel, the 80 staged u columns as two 32-column chunks (K = 32 MFMA) and one
//     16-column chunk (K = 16 MFMA):
//         D[a][b] += sum_k A[a][k] B[k][b],   A[a][k] = u[y + 2 pad - a][32 X + k],   B[k][b] = e'[y][32 X + k - 2 pad + b]
//     A = 16 lane rows of the u planes (ds_read_b128 / b64), B = sliding window of the e' row (five / three dword pairs +
//     v_alignbit, as in ics_gradk_mfma.hip), three MFMAs (hi*hi, hi*lo, lo*hi) per chunk; wave w owns rows 16w..16w+15.
//   * accumulators are folded into fp32 totals per tile and channel with the exact inverse scales; one partial block per
//     workgroup, reduced in double by k_gradk_reduce (ics_kernels.hip), deterministic.
#include "ics_kernels.h"
#include "ics_image_acc.h"
#include <type_traits>

#ifndef ICS_FUSED_INTERLEAVE
#define ICS_FUSED_INTERLEAVE 1
#endif

// Measurement hooks (phase timing, timeline, ablations): empty in the library; the harness builds of tools/bench_synth_gradk.hip define
// ICS_FUSED_PROBES and get their bodies from tools/ics_synth_gradk_probe.h.
#ifdef ICS_FUSED_PROBES
#include "tools/ics_synth_gradk_probe.h"
#else
#define FTICK_INIT
#define FTICK(i)
#define FTICK_FLUSH
#define ICS_FUSED_ABL(mask) 0
#endif
#ifndef ICS_FUSED_GK_INTERLEAVE
#define ICS_FUSED_GK_INTERLEAVE 1
#endif
// Fair shares for the workgroups of one CU.  The walk is static (the partial sums of a workgroup must not depend on timing), every
// workgroup has the same number of tiles, and the CU's arbiter serves the OLDEST wave first: of the two workgroups that share a CU (blocks b
// and b + CUs) the first-dispatched one walked its 8 tiles of a 4096^2 frame in 199 us, the other needed 256 -- 37 us per tile beside its
// mate, 20 alone on a half-empty CU for the last 57 us (phase timeline, tools/bench_synth_gradk.hip -DICS_FUSED_TRACE,
// scripts/dbg/trace_teams.py).  Priority now alternates between the mates in slices of 2^ICS_FUSED_SLICE ticks of the 100 MHz wall clock
// (ics_prio_turn, ics_common.h: s_setprio behind the barriers of the tile).  Scheduling only: results are bit-identical.
// 4096^2, 15x15, 64-row form (tools/bench_synth_gradk.hip, sustained): 0.2744 -> 0.2616 ms with slices of 2^10 ticks (2^7: 0.262, 2^12: 0.262,
// 2^14: 0.266; the younger mate always first: 0.275); the mates now end at 219 / 233 us instead of 199 / 256.
#ifndef ICS_FUSED_SLICE
#define ICS_FUSED_SLICE 10   /* 0: off */
#endif
// (the 32-row form, three mates: every slice length measured slower than none, 0.2646 -> 0.267 ... 0.273 ms; it keeps the arbiter's order)
#define ICS_FUSED_TURN(team, nteams) ics_prio_turn(ICS_FUSED_SLICE, team, nteams)
#ifndef ICS_FUSED_PRIO
#define ICS_FUSED_PRIO 0
#endif
#ifndef ICS_FUSED_F01
#define ICS_FUSED_F01 0   /* image operand of channels 0 and 1 in one dwordx2 request (0: one dword request per channel) */
#endif

#ifndef ICS_FUSED_WSPLIT
#define ICS_FUSED_WSPLIT 0   /* measured: stand-alone harness 0.2749 -> 0.2622 ms (-4.6 %), inside the iteration 0.2739 -> 0.2767 (same box, two
                                libraries alternating): the restaging of a channel's rows from the global table sits on the critical path between the
                                two barriers there.  Built, correct (every fused-kernel test with the 64-row form forced), default off. */
#endif

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
typedef uint32_t u2 __attribute__((ext_vector_type(2)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));

// row classes of the u planes (rows c, c+4, ... contiguous): sizes and padded byte offsets.  The gradient reads 16 CONSECUTIVE rows
// per A fragment (4 rows of each class) and always starts on an even row (one fragment serves the residual rows 2p and 2p + 1,
// see gradk_phase); the class bases are padded so that those reads spread over the banks -- brute-force search over the offsets
// modulo 256 with the ds_read_b128 lane groups of gfx950: 6 LDS cycles on average over the two row phases (8 unpadded; 4 =
// conflict-free, which the y-mod-4 grouping the convolution needs does not allow for consecutive rows)
template <int K>
struct FRows {
  static constexpr int LROWS = 64 + K - 1, ROWB = 160;
  static constexpr int cls_rows(int c) { return (LROWS - c + 3) / 4; }
  static constexpr int want(int c) { return c == 0 ? 0 : (c == 1 ? 64 : (c == 2 ? 32 : 96)); }
  static constexpr int cls_off(int c) {
    if (c == 0) return 0;
    int off = cls_off(c - 1) + cls_rows(c - 1) * ROWB;
    while (off % 256 != want(c)) off += 16;
    return off;
  }
};

template <int K>
struct FCfg {
  static constexpr int PAD = K / 2;
  static constexpr int TH = 64, TW = 64;
  static constexpr int NW = 4, NT = 64 * NW;
  static constexpr int LROWS = TH + K - 1;           // staged u rows
  static constexpr int LCOLS = TW + 16;              // staged u columns [x0 - PAD, x0 - PAD + 80)
  static constexpr int ROWB = 2 * LCOLS;             // 160 bytes per plane row
  static constexpr int OFF0 = 0, OFF1 = FRows<K>::cls_off(1), OFF2 = FRows<K>::cls_off(2), OFF3 = FRows<K>::cls_off(3);
  static constexpr int cls_off(int c) { return c == 0 ? OFF0 : (c == 1 ? OFF1 : (c == 2 ? OFF2 : OFF3)); }
  static constexpr int PLANE = ((OFF3 + FRows<K>::cls_rows(3) * ROWB + 32 + 15) / 16) * 16;   // + 32: the third chunk over-reads a row
  static constexpr int UOFF = 0;                       // [buffer][hi/lo] planes
  static constexpr int EROWB = 4 * LCOLS;              // 320 bytes: (hi, lo) dword pairs of 80 halves, x = -8 .. 71
  static constexpr int EOFF = 4 * PLANE;
  static constexpr int EBYTES = TH * EROWB + 64;       // + slack: the five-pair read of the last row
  static constexpr int SCR = EOFF + EBYTES;            // 256 bytes of floats
  static constexpr int WROWB = (2 * (K + 17) + 3) & ~3;
  static constexpr int WZERO = (K + 7) / 2;
  static constexpr int WLDS = 3 * K * 2 * WROWB;
  static constexpr int WOFF = SCR + 256;
  // ICS_FUSED_WSPLIT (round 4): the weight rows of ONE channel at a time, as four plain rows per kernel row -- hi, lo, hi moved up one
  // half, lo moved up one half -- so that a lane finds its 8 halves dword-aligned in the copy of its parity and reads them with two
  // ds_read2_b32 per split term straight into the operand registers: 4 LDS instructions and no funnel shift per B fragment instead of
  // 5 + 8 (ics_conv_mfma.hip, MCfg::WSPLIT).  All three channels in that form would need 11.5 KB where 5.8 KB are free; one channel
  // (3.8 KB) is restaged from the global table between the two barriers every channel already has.
  static constexpr bool WSPLIT = ICS_FUSED_WSPLIT != 0;
  static constexpr int WLDS_USED = WSPLIT ? K * 4 * WROWB : WLDS;
  static constexpr size_t LDS_BYTES = WOFF + WLDS_USED;
  static constexpr int NQ = K + 3;
  static constexpr int XG = LCOLS / 4;
  static constexpr int NTASK = LROWS * XG;
  static constexpr int NIT = (NTASK + NT - 1) / NT;
  static_assert(K >= 3 && K <= 15 && (K & 1), "one 32-wide MFMA window per column block, one 16-tap block");
  static_assert(2 * LDS_BYTES <= 160 * 1024, "two workgroups per CU");
};

#define ICS_BUF_WORD3 0x00020000  /* gfx9 raw buffer: DATA_FORMAT = 32 */
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7FFFFFFF, ICS_BUF_WORD3);
}

// power-of-two scale that brings a maximum magnitude m into [2^14, 2^15); 1 for m = 0 / Inf / NaN.  `inv` is the exact inverse.
__device__ __forceinline__ void pow2_scale(float m, float& s, float& inv) {
  const uint32_t e = (__float_as_uint(m) >> 23) & 0xFFu;
  uint32_t sb = 127u;
  if (m > 0.f && e != 255u) { sb = 268u - e; sb = sb > 240u ? 240u : sb; }
  s = __uint_as_float(sb << 23);
  inv = __uint_as_float((254u - sb) << 23);
}

__device__ __forceinline__ int opaque(int x) { asm volatile("" : "+v"(x)); return x; }

// workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every outstanding global load (vmcnt(0)):
// with the image operand or the next tile's rows in flight it stalled the whole workgroup for an HBM round trip.
__device__ __forceinline__ void lds_barrier() {
  if (ICS_FUSED_ABL(128)) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
// harness hooks (ICS_FUSED_ABL, a constant 0 in the library): the matrix instruction, or a stand-in that keeps its operands alive; the funnel shift or its first operand
__device__ __forceinline__ f4 f_mfma32(h8 a, h8 b, f4 c) {
  if (ICS_FUSED_ABL(64)) { asm volatile("" :: "v"(a), "v"(b)); return c; }
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f4 f_mfma16(h4 a, h4 b, f4 c) {
  if (ICS_FUSED_ABL(64)) { asm volatile("" :: "v"(a), "v"(b)); return c; }
  return __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ uint32_t f_align(uint32_t hi, uint32_t lo, uint32_t sh) {
  if (ICS_FUSED_ABL(32)) return lo;
  return __builtin_amdgcn_alignbit(hi, lo, sh);
}

template <typename C>
__device__ __forceinline__ void load_raw(f32x4u (&v)[C::NIT][3], __amdgpu_buffer_rsrc_t rs, int soff, int tid, int pitch) {
#pragma unroll
  for (int k = 0; k < C::NIT; ++k) {
    int t = tid + k * C::NT;
    t = t < C::NTASK ? t : C::NTASK - 1;
    const int row = t / C::XG, xg = t - row * C::XG;
    const int toff = 4 * (row * pitch + 12 * xg);
#pragma unroll
    for (int h = 0; h < 3; ++h) v[k][h] = __builtin_bit_cast(f32x4u, __builtin_amdgcn_raw_buffer_load_b128(rs, toff + 16 * h, soff, 0));
  }
}

// one channel of the staged rows -> (hi, lo) fp16 planes, rows grouped by y mod 4
template <typename C, int CH>
__device__ __forceinline__ void convert_channel(const f32x4u (&raw)[C::NIT][3], float s_x, unsigned char* plane, int tid) {
#pragma unroll
  for (int k = 0; k < C::NIT; ++k) {
    const int t = tid + k * C::NT;
    if (t < C::NTASK) {
      const int row = t / C::XG, xg = t - row * C::XG;
      const int rc = row & 3;
      const int coff = C::cls_off(rc) + (row >> 2) * C::ROWB;
      unsigned char* dst = plane + coff + 8 * xg;
      h4 hi, lo;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const int idx = 3 * p + CH;
        const float x = raw[k][idx >> 2][idx & 3] * s_x;
        const _Float16 xh = (_Float16)x;
        hi[p] = xh;
        lo[p] = (_Float16)(x - (float)xh);
      }
      *reinterpret_cast<h4*>(dst) = hi;
      *reinterpret_cast<h4*>(dst + C::PLANE) = lo;
    }
  }
}

template <int K, bool ACC>   // ACC: the image operand comes from the accumulator-order copy a.facc (ics_image_acc.h)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_synth_gradk(IcsFusedArgs a) {
  using C = FCfg<K>;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  float* fscr = reinterpret_cast<float*>(lds + C::SCR);
  const int tid = threadIdx.x;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, li = lane & 15, lg = lane >> 4;
  const int pitch = a.g.pitch;

  // persistent tile walk over the M x N interior (tiles start at (PAD, PAD)), one contiguous band of tiles per XCD
  constexpr int TORG = C::PAD;
  const int tpr = (a.g.N + C::TW - 1) / C::TW;
  const int ntiles = tpr * ((a.g.M + C::TH - 1) / C::TH);
  const int nb = (int)gridDim.x < 8 ? (int)gridDim.x : 8;
  const int xcd = blockIdx.x % nb, kx = blockIdx.x / nb;
  const int nx = ((int)gridDim.x + nb - 1 - xcd) / nb;
  const int band0 = ics_band_begin(ntiles, (int)gridDim.x, nb, xcd), band1 = ics_band_begin(ntiles, (int)gridDim.x, nb, xcd + 1);
  int tile = band0 + kx;

  f4 tot[3];
  float carry[3] = {0.f, 0.f, 0.f};   // register 0 of the even-row blocks: belongs to tap row 4 lg - 1, i.e. to the lane 16 below (gradk_phase)
#pragma unroll
  for (int c = 0; c < 3; ++c) tot[c] = (f4){0.f, 0.f, 0.f, 0.f};

  // LDS: everything zero once (class padding, over-read slack and the zero columns of the e' rows are never written again),
  // then the weight rows (the global table is the LDS image)
  {
    u4* z = reinterpret_cast<u4*>(lds);
    for (int i = tid; i < C::WOFF / 16; i += C::NT) z[i] = (u4){0u, 0u, 0u, 0u};
    if (!C::WSPLIT) {
      uint32_t* ldsW = reinterpret_cast<uint32_t*>(lds + C::WOFF);
      const uint32_t* tab = reinterpret_cast<const uint32_t*>(a.bt);
      for (int i = tid; i < C::WLDS / 4; i += C::NT) ldsW[i] = tab[i];
    }
  }
  const float inv_w = *reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(a.bt) + C::WLDS);
  // WSPLIT: the weight rows of channel `ch` from the global table (hi dword d at 2d, lo at 2d + 1 of a (c, a) block) into the four plain
  // rows; one dword position per thread (K * WROWB / 4 <= 256), requested (wq_issue) ahead of the work it hides behind, stored (wq_store)
  // before the barrier that precedes the convolution of that channel
  typedef uint32_t wq2 __attribute__((ext_vector_type(2)));
  wq2 wq_a = {0u, 0u}, wq_b = {0u, 0u};
  auto wq_issue = [&](int ch) {
    if (!C::WSPLIT) return;
    constexpr int RD = C::WROWB / 4;
    static_assert(!C::WSPLIT || K * RD <= C::NT, "one dword position per thread");
    const int t0 = opaque(tid);
    const int i = t0 < K * RD ? t0 : K * RD - 1;
    const int ar = i / RD, d = i - ar * RD;
    const wq2* src = reinterpret_cast<const wq2*>(reinterpret_cast<const uint32_t*>(a.bt) + (ch * K + ar) * 2 * RD) + d;
    wq_a = src[0];
    wq_b = d + 1 < RD ? src[1] : (wq2){0u, 0u};
  };
  auto wq_store = [&]() {
    if (!C::WSPLIT) return;
    constexpr int RD = C::WROWB / 4;
    const int i = opaque(tid);
    if (i < K * RD) {
      const int ar = i / RD, d = i - ar * RD;
      uint32_t* dst = reinterpret_cast<uint32_t*>(lds + C::WOFF) + ar * 4 * RD + d;
      dst[0] = wq_a.x; dst[RD] = wq_a.y;
      dst[2 * RD] = __builtin_amdgcn_alignbit(wq_b.x, wq_a.x, 16); dst[3 * RD] = __builtin_amdgcn_alignbit(wq_b.y, wq_a.y, 16);
    }
  };
  wq_issue(0);
  wq_store();

  // ---- lane constants --------------------------------------------------------------------------------------------
  typedef const __attribute__((address_space(3))) uint32_t* lds_u32p;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_u32p)(lds);
  // convolution, B operand (weight rows): see ics_conv_mfma.hip
  uint32_t wa0, wsh;
  {
    const int bo = 8 * lg - li + 15;
    const bool bzero = bo < 8 || bo > K + 14;
    wsh = bzero ? 0u : (uint32_t)(bo & 1) * 16u;
    if (C::WSPLIT) wa0 = lds0 + (uint32_t)C::WOFF + 4u * (uint32_t)(bzero ? C::WZERO : ((bo & 1) * 2 * (C::WROWB / 4) + ((bo - 8 - (bo & 1)) >> 1)));
    else wa0 = lds0 + (uint32_t)C::WOFF + 8u * (uint32_t)(bzero ? C::WZERO : ((bo - 8) >> 1));
    asm volatile("" : "+v"(wa0));
  }
  // convolution, A operand: lane row li of column block wv, 8 halves at 16 wv + 8 lg
  const uint32_t conv_a = lds0 + (uint32_t)(C::UOFF + li * C::ROWB + (16 * wv + 8 * lg) * 2);
  // gradient, A operand.  ONE fragment of 16 consecutive u rows serves TWO residual rows: for the pair (y, y + 1), y = 16 wv + 2 p,
  // lane row m reads u row (y + 1) + 2 pad - m of the staged block; against the residual row y + 1 that is tap a = m (block D1),
  // against the residual row y it is tap a = m - 1 (block D0: its row 0 pairs with nothing and is dropped, rows 1 .. 15 are taps
  // 0 .. 14 -- K <= 15 makes the two tap ranges fit one fragment).  The A reads, the conflicted half of this phase's LDS traffic,
  // are halved.  8 halves at 8 lg of each 32-column chunk; pairs alternate between two row phases (2 p mod 4), each one row further
  // inside its classes every second pair.  Lanes beyond the taps (m > 2 pad + 1) repeat the last valid row: their blocks are never read.
  uint32_t ga[2];
  {
    const int mm = li < 2 * C::PAD + 1 ? li : 2 * C::PAD + 1;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r = 2 * j + 2 * C::PAD + 1 - mm;                // >= 0
      const int rc = r & 3;
      const int coff = C::cls_off(rc) + ((r >> 2) + 4 * wv) * C::ROWB;
      ga[j] = lds0 + (uint32_t)(C::UOFF + coff + 16 * lg);
    }
  }
  // gradient, B operand: lane column li <-> tap b; in chunk X its 8 halves start at e' column s = 32 X + 8 lg + b - 2 pad.
  // Windows entirely left / right of the tile's 64 columns are moved onto the zero columns of the row.
  // The third chunk (columns 64 .. 79) is 16 wide: 4 halves per lane, u columns 64 + 4 lg .., e' window at 64 + 4 lg + b - 2 pad.
  uint32_t gb[3], gsh[3];
  const uint32_t ga2 = (uint32_t)(8 * lg);
  {
    const int tb = li < K ? li : K - 1;
#pragma unroll
    for (int X = 0; X < 3; ++X) {
      int s = X < 2 ? 32 * X + 8 * lg + tb - 2 * C::PAD : 64 + 4 * lg + tb - 2 * C::PAD;
      s = s <= -8 ? -8 : (s >= 64 ? 64 : s);
      gsh[X] = (uint32_t)(s & 1) * 16u;
      gb[X] = lds0 + (uint32_t)(C::EOFF + 16 * wv * C::EROWB + 8 * ((s + 8) >> 1));
    }
  }
  // e' planes, store side: value (row t + 16 lg + 4 r, column 16 wv + li); lanes li, li ^ 1 share a dword pair
  const uint32_t ew = lds0 + (uint32_t)(C::EOFF + 16 * lg * C::EROWB + 8 * ((16 * wv + li + 8) >> 1) + 4 * (li & 1));

  const ptrdiff_t orgoff = (ptrdiff_t)a.g.ay * pitch + 3 * a.g.ax;
  const __amdgpu_buffer_rsrc_t rs_in = make_rsrc(a.u - orgoff);
  const __amdgpu_buffer_rsrc_t rs_f = make_rsrc(a.f);
  const __amdgpu_buffer_rsrc_t rs_o = make_rsrc(a.e_out);

  f32x4u raw[C::NIT][3];
  if (tile < band1) {
    const int tyi = tile / tpr, txi = tile - tyi * tpr;
    load_raw<C>(raw, rs_in, 4 * ((a.g.ay + TORG + tyi * C::TH - C::PAD) * pitch + 3 * (a.g.ax + TORG + txi * C::TW - C::PAD)), tid, pitch);
  }
  __syncthreads();   // LDS initialised
  FTICK_INIT;
  constexpr int MATES = 2;                                                      // workgroups per CU
  const int team = (int)gridDim.x >= MATES ? (int)blockIdx.x / ((int)gridDim.x / MATES) % MATES : 0;

#pragma unroll 1
  for (; tile < band1; tile += nx) {
    const int tyi = tile / tpr, txi = tile - tyi * tpr;
    const int x0 = TORG + txi * C::TW, y0 = TORG + tyi * C::TH;
    const bool store_e = a.store_all || (y0 < a.wy1 && y0 + C::TH > a.wy0 && x0 < a.wx1 && x0 + C::TW > a.wx0);   // wave-uniform

    // ---- per-tile power-of-two scale of u (all three channels) ---------------------------------------------------
    float s_x, inv_x;
    {
      float m = 0.f;
#pragma unroll
      for (int k = 0; k < C::NIT; ++k)
#pragma unroll
        for (int h = 0; h < 3; ++h)
#pragma unroll
          for (int e = 0; e < 4; ++e) m = __builtin_fmaxf(m, __builtin_fabsf(raw[k][h][e]));
      m = ics_wave_max_f32(m);
      if (lane == 0) fscr[wv] = m;
      lds_barrier();     // S0: also orders the previous tile's last gradient phase before the planes are rewritten
      ICS_FUSED_TURN(team, MATES);
#pragma unroll
      for (int w = 0; w < C::NW; ++w) m = __builtin_fmaxf(m, fscr[w]);
      pow2_scale(__builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, m))), s_x, inv_x);
    }
    FTICK(0);
    const float sc = inv_w * inv_x;

    // lane part of the epilogue addresses (image operand, optional e' store): pixel column 16 wv + li, rows 16 lg + ...
    const int tide = opaque(tid);
    const int eli = tide & 15, elg = (tide >> 4) & 3;
    const int colx = x0 + 16 * wv + eli;
    const int voff = 4 * (16 * elg * pitch + 3 * eli);
    const int sb = 4 * (y0 * pitch + 3 * (x0 + 16 * wv));

    // image operand.  With the accumulator-order copy (a.facc, ics_image_acc.h): four 16-byte loads per channel, requested right
    // before the channel's convolution and consumed behind it.  Without (tv_mode 1 rewrites the image every iteration): channels 0
    // and 1 arrive together from the HWC frame (16 dwordx2 requests instead of 32 dword requests per lane: the TA processes a
    // request per instruction, and 48 stride-12 dword requests per lane and tile cost 0.037 ms of the 0.29), channel 1 waits in 16
    // registers across gradk(0) and conv(1); channel 2 is requested on its own before conv(2)
    uint32_t fop[4][4], fop1[ACC ? 1 : 4][4];
    constexpr bool use_acc = ACC;
    const __amdgpu_buffer_rsrc_t rs_acc = make_rsrc(a.facc);
    const int acc_voff = 16 * (tide & 63);
    const int acc_sb = (tile * 4 + wv) * (3 * 4 * 1024);          // bytes: [tile][cb][ch][t][lane] float4
    auto load_acc = [&](int ch) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const u4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_acc, acc_voff, acc_sb + (ch * 4 + t) * 1024, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) fop[t][r] = v[r];
      }
    };
    auto load_f01 = [&]() {
      if constexpr (ACC) return;
      else if (ICS_FUSED_ABL(16)) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) { fop[t][r] = 0u; fop1[t][r] = 0u; }
        return;
      }
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const u2 v = __builtin_amdgcn_raw_buffer_load_b64(rs_f, voff, sb + 4 * (t + 4 * r) * pitch, 0);
          fop[t][r] = v.x; fop1[t][r] = v.y;
        }
    };
    auto take_f1 = [&]() {
      if constexpr (!ACC) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) fop[t][r] = fop1[t][r];
      }
    };
    auto load_f = [&](int ch) {
      if (ICS_FUSED_ABL(16)) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) fop[t][r] = 0u;
        return;
      }
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          fop[t][r] = __builtin_amdgcn_raw_buffer_load_b32(rs_f, voff + 4 * ch, sb + 4 * (t + 4 * r) * pitch, 0);
    };

    f4 acc[4];
    // ---- Toeplitz convolution of one channel from plane buffer (ch & 1): ics_conv_mfma.hip, one channel -------------
    auto conv_phase = [&](auto chc) {
      constexpr int ch = decltype(chc)::value;
      constexpr uint32_t PB = (uint32_t)((ch & 1) * 2 * C::PLANE);
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = (f4){0.f, 0.f, 0.f, 0.f};
      if (ICS_FUSED_ABL(2)) return;
      typedef const __attribute__((address_space(3))) h8* lds_h8p;
      typedef const volatile __attribute__((address_space(3))) u2* lds_vu2p;
      uint32_t wb = wa0; asm volatile("" : "+v"(wb));
      uint32_t ca = conv_a; asm volatile("" : "+v"(ca));
      h8 Bh[K], Bl[K];
      u2 rawB[5];
      auto issueB = [&](int ka) {
        if constexpr (C::WSPLIT) {   // (inline asm: as C++ loads the pairs are merged into ds_read2_b64 at 4-byte alignment, which gfx950 executes very slowly)
          const uint32_t ad = wb + (uint32_t)(ka * 4 * C::WROWB);
          constexpr int RD = C::WROWB / 4;
          u2 h01, h23, l01, l23;
          asm volatile("ds_read2_b32 %0, %1 offset1:1" : "=v"(h01) : "v"(ad));
          asm volatile("ds_read2_b32 %0, %1 offset0:2 offset1:3" : "=v"(h23) : "v"(ad));
          asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(l01) : "v"(ad), "n"(RD), "n"(RD + 1));
          asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(l23) : "v"(ad), "n"(RD + 2), "n"(RD + 3));
          Bh[ka] = __builtin_bit_cast(h8, (u4){h01.x, h01.y, h23.x, h23.y});
          Bl[ka] = __builtin_bit_cast(h8, (u4){l01.x, l01.y, l23.x, l23.y});
          return;
        }
        const lds_vu2p r = reinterpret_cast<lds_vu2p>(wb + (uint32_t)((ch * K + ka) * 2 * C::WROWB));
#pragma unroll
        for (int d = 0; d < 5; ++d) rawB[d] = r[d];
      };
      auto finishB = [&](int ka) {
        if constexpr (C::WSPLIT) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); return; }   // (asm results are not tracked by the compiler's s_waitcnt insertion)
        const u2* d = rawB;
        const u4 wh = {f_align(d[1].x, d[0].x, wsh), f_align(d[2].x, d[1].x, wsh),
                       f_align(d[3].x, d[2].x, wsh), f_align(d[4].x, d[3].x, wsh)};
        const u4 wl = {f_align(d[1].y, d[0].y, wsh), f_align(d[2].y, d[1].y, wsh),
                       f_align(d[3].y, d[2].y, wsh), f_align(d[4].y, d[3].y, wsh)};
        Bh[ka] = __builtin_bit_cast(h8, wh);
        Bl[ka] = __builtin_bit_cast(h8, wl);
      };
      issueB(0);
      h8 Ah = *reinterpret_cast<lds_h8p>(ca + PB), Al = *reinterpret_cast<lds_h8p>(ca + PB + C::PLANE);
      finishB(0);
      __builtin_amdgcn_sched_barrier(0);
      if (ICS_FUSED_PRIO & 2) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int q = 0; q < C::NQ; ++q) {
        h8 Nh = Ah, Nl = Al;
        if (q + 1 < C::NQ) {
            const uint32_t off = (uint32_t)(C::cls_off((q + 1) & 3) + ((q + 1) >> 2) * C::ROWB);
          Nh = *reinterpret_cast<lds_h8p>(ca + PB + off);
          Nl = *reinterpret_cast<lds_h8p>(ca + PB + C::PLANE + off);
        }
        if (q + 1 < K) issueB(q + 1);
#pragma unroll
        for (int term = 0; term < 3; ++term)
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const int ka = q - t;
            if (ka < 0 || ka >= K) continue;
            acc[t] = f_mfma32(term == 2 ? Al : Ah, term == 1 ? Bl[ka] : Bh[ka], acc[t]);
          }
        if (q + 1 < K) finishB(q + 1);
        Ah = Nh; Al = Nl;
        if (ICS_FUSED_INTERLEAVE) {
          int nt = 0;
#pragma unroll
          for (int t = 0; t < 4; ++t) nt += (q - t >= 0 && q - t < K) ? 1 : 0;
          const int nm = 3 * nt;
          const int nr = ((q + 1 < C::NQ) ? 2 : 0) + ((q + 1 < K) ? (C::WSPLIT ? 4 : 5) : 0);
          const int nv = (q + 1 < K && !C::WSPLIT) ? 8 : 0;
          const int tail = nv ? (nm > 4 ? 4 : nm) : 0;
          const int head = nm - tail;
#pragma unroll
          for (int i = 0; i < (head > nr ? head : nr); ++i) {
            if (i < head) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (i < nr) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          }
#pragma unroll
          for (int i = 0; i < tail; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
#pragma unroll
            for (int j = 0; j < (nv / 2 + tail - 1) / tail; ++j) __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    };

    // ---- e'(ch) = conv - image (0 outside the M x N interior), kept in `acc`; returns the lane's max |e'| ---------------
    auto residual = [&](int ch) -> float {
      float m = 0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int y = y0 + t + 16 * elg + 4 * r;
          const bool in = y < C::PAD + a.g.M && colx < C::PAD + a.g.N;   // (tiles start at (PAD, PAD))
          const float e = in ? __fsub_rn(acc[t][r] * sc, __uint_as_float(fop[t][r])) : 0.f;
          acc[t][r] = e;
          m = __builtin_fmaxf(m, __builtin_fabsf(e));
          if (store_e && in) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(e), rs_o, voff + 4 * ch, sb + 4 * (t + 4 * r) * pitch, 0);
        }
      return ics_wave_max_f32(m);
    };

    // ---- e' -> fp16 (hi, lo) planes: a lane packs (hi | lo << 16), swaps with its column neighbour and stores one dword --
    auto write_e = [&](float s_e) {
      typedef __attribute__((address_space(3))) uint32_t* lds_wp;
      if (ICS_FUSED_ABL(4)) return;
      const bool odd = (li & 1) != 0;
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float x = acc[t][r] * s_e;
          const _Float16 xh = (_Float16)x;
          const _Float16 xl = (_Float16)(x - (float)xh);
          const uint32_t P = (uint32_t)__builtin_bit_cast(unsigned short, xh) | ((uint32_t)__builtin_bit_cast(unsigned short, xl) << 16);
          const uint32_t Q = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)P, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
          // even column: hi dword = (own hi, neighbour hi); odd column: lo dword = (neighbour lo, own lo)
          const uint32_t w = odd ? ((Q >> 16) | (P & 0xFFFF0000u)) : ((P & 0xFFFFu) | (Q << 16));
          *reinterpret_cast<lds_wp>(ew + (uint32_t)((t + 4 * r) * C::EROWB)) = w;
        }
    };

    // ---- PSF gradient of one channel: 16 residual rows of this wave x (2 chunks of 32 + 1 chunk of 16 columns) x 3 split terms
    // The 80 staged u columns are two 32-column chunks (v_mfma_f32_16x16x32_f16) and one 16-column chunk (the K = 16 form,
    // v_mfma_f32_16x16x16_f16: 8-byte A fragments, a 4-half e' window from three dword pairs).  Same matrix-pipe time as a
    // third 32-column chunk (measured: both forms issue at the same rate), but 10 instead of 24 LDS cycles, and this phase is
    // LDS-bound.
    // Software pipeline, one row deep: the LDS operands of row i + 1 are requested in the shadows of the first MFMAs of row i
    // (the e' windows first: they still need their funnel shifts, which go behind the later MFMAs of the same row).  The reads
    // are volatile: plain loads were merged into ds_read2_b64 (8 LDS cycles for two 8-byte reads instead of 2 + 2) and sunk to
    // their first use.
    auto gradk_phase = [&](auto chc, float scale) {
      constexpr int ch = decltype(chc)::value;
      constexpr uint32_t PB = (uint32_t)((ch & 1) * 2 * C::PLANE);
      typedef const volatile __attribute__((address_space(3))) u4* lds_vu4p;
      typedef const volatile __attribute__((address_space(3))) u2* lds_vu2p;
      // [residual row parity][2]: the three column chunks and three split terms of a row are nine summands of the same block; they
      // alternate between two accumulators so that no MFMA follows one on the same registers (three would cost 8 more VGPRs)
      f4 g[2][2];
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int X = 0; X < 2; ++X) g[h][X] = (f4){0.f, 0.f, 0.f, 0.f};
      if (ICS_FUSED_ABL(1)) { tot[ch][0] += scale; return; }
      uint32_t gav[2], gbv[3];
#pragma unroll
      for (int j = 0; j < 2; ++j) { gav[j] = ga[j]; asm volatile("" : "+v"(gav[j])); }
#pragma unroll
      for (int X = 0; X < 3; ++X) { gbv[X] = gb[X]; asm volatile("" : "+v"(gbv[X])); }
      u4 Ah[2], Al[2], nAh[2], nAl[2];
      u2 A2h, A2l, nA2h, nA2l;           // third chunk: 4 halves per lane
      u2 rB[2][5], rB2[3];
      h8 Bh[2], Bl[2];
      h4 B2h, B2l;
      // residual row i: its e' windows always; the u fragment of the pair only on even rows
      auto issue = [&](int i) {
#pragma unroll
        for (int X = 0; X < 2; ++X) {
          const lds_vu2p ep = reinterpret_cast<lds_vu2p>(gbv[X] + (uint32_t)(i * C::EROWB));
#pragma unroll
          for (int k = 0; k < 5; ++k) rB[X][k] = ep[k];
        }
        {
          const lds_vu2p ep = reinterpret_cast<lds_vu2p>(gbv[2] + (uint32_t)(i * C::EROWB));
#pragma unroll
          for (int k = 0; k < 3; ++k) rB2[k] = ep[k];
        }
        if ((i & 1) == 0) {
          const uint32_t ar = gav[(i >> 1) & 1] + PB + (uint32_t)((i >> 2) * C::ROWB);
#pragma unroll
          for (int X = 0; X < 2; ++X) {
            nAh[X] = *reinterpret_cast<lds_vu4p>(ar + 64 * X);
            nAl[X] = *reinterpret_cast<lds_vu4p>(ar + C::PLANE + 64 * X);
          }
          // third chunk: columns 64 + 4 lg .. + 3 (the lane address carries 16 lg: back by 8 lg)
          nA2h = *reinterpret_cast<lds_vu2p>(ar + 128 - ga2);
          nA2l = *reinterpret_cast<lds_vu2p>(ar + C::PLANE + 128 - ga2);
        }
      };
      auto finish = [&]() {
#pragma unroll
        for (int X = 0; X < 2; ++X) {
          const u2* d = rB[X];
          const u4 wh = {f_align(d[1].x, d[0].x, gsh[X]), f_align(d[2].x, d[1].x, gsh[X]),
                         f_align(d[3].x, d[2].x, gsh[X]), f_align(d[4].x, d[3].x, gsh[X])};
          const u4 wl = {f_align(d[1].y, d[0].y, gsh[X]), f_align(d[2].y, d[1].y, gsh[X]),
                         f_align(d[3].y, d[2].y, gsh[X]), f_align(d[4].y, d[3].y, gsh[X])};
          Bh[X] = __builtin_bit_cast(h8, wh);
          Bl[X] = __builtin_bit_cast(h8, wl);
        }
        const u2 wh2 = {f_align(rB2[1].x, rB2[0].x, gsh[2]), f_align(rB2[2].x, rB2[1].x, gsh[2])};
        const u2 wl2 = {f_align(rB2[1].y, rB2[0].y, gsh[2]), f_align(rB2[2].y, rB2[1].y, gsh[2])};
        B2h = __builtin_bit_cast(h4, wh2);
        B2l = __builtin_bit_cast(h4, wl2);
      };
      issue(0);
      finish();
#pragma unroll
      for (int X = 0; X < 2; ++X) { Ah[X] = nAh[X]; Al[X] = nAl[X]; }
      A2h = nA2h; A2l = nA2l;
      __builtin_amdgcn_sched_barrier(0);
      if (ICS_FUSED_PRIO & 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int hp = i & 1;
        h8 cBh[2], cBl[2];
#pragma unroll
        for (int X = 0; X < 2; ++X) { cBh[X] = Bh[X]; cBl[X] = Bl[X]; }
        const h4 cB2h = B2h, cB2l = B2l;
        if (i + 1 < 16) issue(i + 1);
#pragma unroll
        for (int term = 0; term < 3; ++term) {
#pragma unroll
          for (int X = 0; X < 2; ++X) {
            const int k = (3 * term + X) & 1;
            g[hp][k] = f_mfma32(__builtin_bit_cast(h8, term == 2 ? Al[X] : Ah[X]), term == 1 ? cBl[X] : cBh[X], g[hp][k]);
          }
          const int k2 = (3 * term + 2) & 1;
          g[hp][k2] = f_mfma16(__builtin_bit_cast(h4, term == 2 ? A2l : A2h), term == 1 ? cB2l : cB2h, g[hp][k2]);
        }
        if (i + 1 < 16) {
          finish();
          if (hp) {   // the next row opens a new pair: its fragment was requested in this step
#pragma unroll
            for (int X = 0; X < 2; ++X) { Ah[X] = nAh[X]; Al[X] = nAl[X]; }
            A2h = nA2h; A2l = nA2l;
          }
        }
        if (ICS_FUSED_GK_INTERLEAVE && i + 1 < 16) {
          // 9 MFMAs; LDS reads of the next row: 13 e' dword pairs, + 4 + 2 u fragments when it opens a pair; 20 funnel shifts
          if (hp) {
#pragma unroll
            for (int k = 0; k < 9; ++k) {
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
              if (k < 6) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
              if (k == 6) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
              if (k >= 4) __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
            }
          } else {
#pragma unroll
            for (int k = 0; k < 9; ++k) {
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
              if (k < 4) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
              if (k == 4) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
              if (k >= 4) __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (ICS_FUSED_PRIO & 1) __builtin_amdgcn_s_setprio(0);
      // block D1 (odd residual rows): row m = tap m.  Block D0 (even rows): row m = tap m - 1, i.e. tap a sits one row further down:
      // row a + 1 = register r + 1 of the same lane, or -- for r = 3 -- register 0 of the lane 16 above (rows 4 lg + r).  That one
      // crosses lanes: it is accumulated apart (`carry`) and joins its tap in the final cross-wave reduction, which goes through LDS anyway.
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float s1 = g[1][0][r] + g[1][1][r];
        const float s0n = r < 3 ? g[0][0][r + 1] + g[0][1][r + 1] : 0.f;
        tot[ch][r] += (s1 + s0n) * scale;
      }
      carry[ch] += (g[0][0][0] + g[0][1][0]) * scale;
    };

    // ================================ the tile ======================================================================
    unsigned char* const up = lds + C::UOFF;
    convert_channel<C, 0>(raw, s_x, up, opaque(tid));
    lds_barrier();                                                     // planes of channel 0 visible
    FTICK(1);
    if (use_acc) load_acc(0); else if (ICS_FUSED_F01) load_f01(); else load_f(0);
    conv_phase(std::integral_constant<int, 0>{});
    FTICK(2);
    float s_e, inv_e, me;

#define ICS_FUSED_CHANNEL(CH)                                                                                           \
    me = residual(CH);                                                                                                  \
    if (lane == 0) fscr[8 + 4 * (CH) + wv] = me;                                                                        \
    FTICK(3);                                                                                                           \
    lds_barrier();     /* tile maximum; every wave is past the previous gradient phase: e' planes and u buffer free */  \
    FTICK(4);                                                                                                           \
    wq_issue(((CH) + 1) % 3);   /* every wave is past conv(CH): the weight rows of the next channel may take its place */          \
    me = __builtin_fmaxf(__builtin_fmaxf(fscr[8 + 4 * (CH)], fscr[9 + 4 * (CH)]), __builtin_fmaxf(fscr[10 + 4 * (CH)], fscr[11 + 4 * (CH)])); \
    pow2_scale(__builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, me))), s_e, inv_e);      \
    write_e(s_e);                                                                                                       \
    wq_store();

    ICS_FUSED_CHANNEL(0)
    if (!(ICS_FUSED_ABL(8))) convert_channel<C, 1>(raw, s_x, up + 2 * C::PLANE, opaque(tid));
    FTICK(5);
    lds_barrier();                                                     // e'(0) and planes(1) visible
    FTICK(6);
    ICS_FUSED_TURN(team, MATES);
    gradk_phase(std::integral_constant<int, 0>{}, inv_x * inv_e);
    FTICK(7);
    if (use_acc) load_acc(1); else if (ICS_FUSED_F01) take_f1(); else load_f(1);
    conv_phase(std::integral_constant<int, 1>{});
    FTICK(2);

    ICS_FUSED_CHANNEL(1)
    if (!(ICS_FUSED_ABL(8))) convert_channel<C, 2>(raw, s_x, up, opaque(tid));
    FTICK(5);
    lds_barrier();                                                     // e'(1) and planes(2) visible
    FTICK(6);
    ICS_FUSED_TURN(team, MATES);
    gradk_phase(std::integral_constant<int, 1>{}, inv_x * inv_e);
    FTICK(7);
    if (use_acc) load_acc(2); else load_f(2);
    // the rows of the next tile: in flight during conv(2), gradk(2) (no vector-memory loads in there; the image operand of
    // channel 2 was requested before them and returns first).  (Spreading the 21 requests of a lane over the steps of conv(2)
    // instead of one burst measured slower: 0.315 vs 0.285 ms.)
    if (tile + nx < band1) {
      const int nt = tile + nx;
      const int nyi = nt / tpr, nxi = nt - nyi * tpr;
      load_raw<C>(raw, rs_in, 4 * ((a.g.ay + TORG + nyi * C::TH - C::PAD) * pitch + 3 * (a.g.ax + TORG + nxi * C::TW - C::PAD)), opaque(tid), pitch);
    }
    __builtin_amdgcn_sched_barrier(0);
    FTICK(8);
    conv_phase(std::integral_constant<int, 2>{});
    FTICK(2);

    ICS_FUSED_CHANNEL(2)
    FTICK(5);
    lds_barrier();                                                     // e'(2) visible
    FTICK(6);
    ICS_FUSED_TURN(team, MATES);
    gradk_phase(std::integral_constant<int, 2>{}, inv_x * inv_e);
    FTICK(7);
#undef ICS_FUSED_CHANNEL
  }

  FTICK_FLUSH;
  // ---- cross-wave reduction (fixed order) and partial write, one channel per pass ------------------------------------
  float* red = reinterpret_cast<float*>(lds);   // [wave][256]: element (row = 4*lg + r, col = li) at [r*64 + lane]; then [wave][64] carries
  float* dst = a.partial + (size_t)blockIdx.x * (3 * 16 * 16);
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wv * 256 + r * 64 + lane] = tot[c][r];
    red[C::NW * 256 + wv * 64 + lane] = carry[c];
    __syncthreads();
    {
      const int v = tid;
      float s = red[v];
#pragma unroll
      for (int w = 1; w < C::NW; ++w) s += red[w * 256 + v];   // fixed order -> deterministic
      const int l = v & 63, r = (v >> 6) & 3;
      if (r == 3 && l < 48) {                                   // tap row 4 lg + 3 also receives row 0 of the lane group above
#pragma unroll
        for (int w = 0; w < C::NW; ++w) s += red[C::NW * 256 + w * 64 + l + 16];
      }
      const int ta = 4 * (l >> 4) + r, tb = l & 15;
      dst[(c * 16 + ta) * 16 + tb] = s;
    }
  }
}


// =====================================================================================================================
// 32-row tiles, THREE workgroups per CU (round 3).  Same arithmetic as k_synth_gradk above, different shape:
//   * tiles of 32 x 64 pixels; fragment rows 2 apart (two accumulator sets per wave and channel, as MCfg<K, 2> of
//     ics_conv_mfma.hip); the u planes hold 32 + K - 1 rows in two row classes (even / odd rows), the class-1 base sits at
//     128 (mod 256) bytes: the gradient's A fragments -- 16 CONSECUTIVE rows, 8 of each class, always starting on an even row --
//     are then conflict-free for the ds_read_b128 lane groups (4 LDS cycles; the four-class layout of the 64-row kernel: 6);
//   * 46 KB of LDS and <= 168 VGPRs -> three resident workgroups per CU (the 64-row kernel: 79 KB, 256 VGPRs, two);
//   * written for thread-level parallelism instead of instruction-level pipelining: the 64-row kernel hides LDS latency inside a
//     wave with one-row-deep software pipelines and copies of the operand registers (130 VGPRs in the gradient loop alone);
//     here a wave requests an operand chunk, waits, shifts, issues its three MFMAs, and the two other waves of the SIMD fill
//     the gaps -- 70 VGPRs in the gradient loop, which is what makes the third workgroup fit.
// The ablation of the 64-row kernel that motivates it (tools/bench_synth_gradk.hip -DICS_FUSED_ABLATE, 4096^2, K = 15): without any
// MFMA the kernel still takes 0.177 of its 0.265 ms -- a wave issues 6.2 k instructions per tile (25 k cycles of its 59 k) and
// stalls for the rest; two waves per SIMD leave that unhidden.
// =====================================================================================================================
#ifndef ICS_FUSED2_PREFETCH
#define ICS_FUSED2_PREFETCH 0   /* measured: 0 (request at the start of the tile, 137 VGPRs) 0.2624 ms, 1 0.2669, 2 (ahead of conv(2), 167 VGPRs + spills) 0.2658 */
#endif
template <int K>
struct FCfg2 {
  static constexpr int PAD = K / 2;
  static constexpr int TH = 32, TW = 64;
  static constexpr int NW = 4, NT = 64 * NW;
  static constexpr int LROWS = TH + K - 1;
  static constexpr int LCOLS = TW + 16;
  static constexpr int ROWB = 2 * LCOLS;                                   // 160
  static constexpr int cls_rows(int c) { return (LROWS - c + 1) / 2; }
  static constexpr int cls1_off() {
    int off = cls_rows(0) * ROWB;
    while (off % 256 != 128) off += 16;
    return off;
  }
  static constexpr int OFF1 = cls1_off();
  static constexpr int cls_off(int c) { return c == 0 ? 0 : OFF1; }
  static constexpr int PLANE = ((OFF1 + cls_rows(1) * ROWB + 32 + 15) / 16) * 16;   // + 32: the third chunk over-reads a row
  static constexpr int UOFF = 0;
  static constexpr int EROWB = 4 * LCOLS;                                  // 320
  static constexpr int EOFF = 4 * PLANE;
  static constexpr int EBYTES = TH * EROWB + 64;
  static constexpr int SCR = EOFF + EBYTES;
  static constexpr int WROWB = (2 * (K + 17) + 3) & ~3;
  static constexpr int WZERO = (K + 7) / 2;
  static constexpr int WLDS = 3 * K * 2 * WROWB;
  static constexpr int WOFF = SCR + 256;
  static constexpr size_t LDS_BYTES = WOFF + WLDS;
  static constexpr int NQ = K + 1;
  static constexpr int XG = LCOLS / 4;
  static constexpr int NTASK = LROWS * XG;
  static constexpr int NIT = (NTASK + NT - 1) / NT;
  static_assert(K >= 3 && K <= 15 && (K & 1), "one 32-wide MFMA window per column block, one 16-tap block");
  static_assert(3 * LDS_BYTES <= 160 * 1024, "three workgroups per CU");
};

template <typename C>
__device__ __forceinline__ void load_raw2(f32x4u (&v)[C::NIT][3], __amdgpu_buffer_rsrc_t rs, int soff, int tid, int pitch) {
#pragma unroll
  for (int k = 0; k < C::NIT; ++k) {
    int t = tid + k * C::NT;
    t = t < C::NTASK ? t : C::NTASK - 1;
    const int row = t / C::XG, xg = t - row * C::XG;
    const int toff = 4 * (row * pitch + 12 * xg);
#pragma unroll
    for (int h = 0; h < 3; ++h) v[k][h] = __builtin_bit_cast(f32x4u, __builtin_amdgcn_raw_buffer_load_b128(rs, toff + 16 * h, soff, 0));
  }
}

// one channel of the staged rows -> (hi, lo) fp16 planes, rows grouped by y mod 2
template <typename C, int CH>
__device__ __forceinline__ void convert_channel2(const f32x4u (&raw)[C::NIT][3], float s_x, unsigned char* plane, int tid) {
#pragma unroll
  for (int k = 0; k < C::NIT; ++k) {
    const int t = tid + k * C::NT;
    if (t < C::NTASK) {
      const int row = t / C::XG, xg = t - row * C::XG;
      unsigned char* dst = plane + ((row & 1) ? C::OFF1 : 0) + (row >> 1) * C::ROWB + 8 * xg;
      h4 hi, lo;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const int idx = 3 * p + CH;
        const float x = raw[k][idx >> 2][idx & 3] * s_x;
        const _Float16 xh = (_Float16)x;
        hi[p] = xh;
        lo[p] = (_Float16)(x - (float)xh);
      }
      *reinterpret_cast<h4*>(dst) = hi;
      *reinterpret_cast<h4*>(dst + C::PLANE) = lo;
    }
  }
}

template <int K, bool ACC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_synth_gradk2(IcsFusedArgs a) {
  using C = FCfg2<K>;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  float* fscr = reinterpret_cast<float*>(lds + C::SCR);
  const int tid = threadIdx.x;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int pitch = a.g.pitch;

  constexpr int TORG = C::PAD;
  const int tpr = (a.g.N + C::TW - 1) / C::TW;
  const int ntiles = tpr * ((a.g.M + C::TH - 1) / C::TH);
  const int nb = (int)gridDim.x < 8 ? (int)gridDim.x : 8;
  const int xcd = blockIdx.x % nb, kx = blockIdx.x / nb;
  const int nx = ((int)gridDim.x + nb - 1 - xcd) / nb;
  const int band0 = ics_band_begin(ntiles, (int)gridDim.x, nb, xcd), band1 = ics_band_begin(ntiles, (int)gridDim.x, nb, xcd + 1);
  int tile = band0 + kx;

  f4 tot[3];
  float carry[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < 3; ++c) tot[c] = (f4){0.f, 0.f, 0.f, 0.f};

  // Prologue: the weight rows (the global table is the LDS image) are requested first, into registers, then the first tile's rows; the
  // LDS is zeroed and the weight rows stored while the tile rows are in flight.  (As zero-fill, a load / store loop for the weights and
  // then the tile rows these were three round trips in series -- what a frame of one tile per workgroup, e.g. deblur_module's 255-px
  // blind windows, spends its time on; as in ics_conv_mfma.hip.)
  constexpr int WPT = (C::WLDS / 4 + C::NT - 1) / C::NT;
  uint32_t wreg[WPT];
  {
    const uint32_t* tab = reinterpret_cast<const uint32_t*>(a.bt);
#pragma unroll
    for (int k = 0; k < WPT; ++k) { const int i = tid + k * C::NT; wreg[k] = i < C::WLDS / 4 ? tab[i] : 0u; }
  }

  typedef const __attribute__((address_space(3))) uint32_t* lds_u32p;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_u32p)(lds);
  const ptrdiff_t orgoff = (ptrdiff_t)a.g.ay * pitch + 3 * a.g.ax;
  const __amdgpu_buffer_rsrc_t rs_in = make_rsrc(a.u - orgoff);
  const __amdgpu_buffer_rsrc_t rs_f = make_rsrc(a.f);
  const __amdgpu_buffer_rsrc_t rs_o = make_rsrc(a.e_out);
  const __amdgpu_buffer_rsrc_t rs_acc = make_rsrc(a.facc);

  f32x4u raw[C::NIT][3];
  if (tile < band1) {
    const int tyi = tile / tpr, txi = tile - tyi * tpr;
    load_raw2<C>(raw, rs_in, 4 * ((a.g.ay + TORG + tyi * C::TH - C::PAD) * pitch + 3 * (a.g.ax + TORG + txi * C::TW - C::PAD)), tid, pitch);
  }
  {
    u4* z = reinterpret_cast<u4*>(lds);
    for (int i = tid; i < C::WOFF / 16; i += C::NT) z[i] = (u4){0u, 0u, 0u, 0u};
    uint32_t* ldsW = reinterpret_cast<uint32_t*>(lds + C::WOFF);
#pragma unroll
    for (int k = 0; k < WPT; ++k) { const int i = tid + k * C::NT; if (i < C::WLDS / 4) ldsW[i] = wreg[k]; }
  }
  const float inv_w = *reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(a.bt) + C::WLDS);
  __syncthreads();   // LDS initialised
  // (no priority slices here: with three workgroups per CU every slice length, and a rotation in which one mate steps back, measured slower
  //  than the arbiter's own order -- ICS_FUSED_TURN above)

#pragma unroll 1
  for (; tile < band1; tile += nx) {
    const int tyi = tile / tpr, txi = tile - tyi * tpr;
    const int x0 = TORG + txi * C::TW, y0 = TORG + tyi * C::TH;
    const bool store_e = a.store_all || (y0 < a.wy1 && y0 + C::TH > a.wy0 && x0 < a.wx1 && x0 + C::TW > a.wx0);   // wave-uniform

    // ---- per-tile power-of-two scale of u (all three channels) ---------------------------------------------------
    float s_x, inv_x;
    {
      float m = 0.f;
#pragma unroll
      for (int k = 0; k < C::NIT; ++k)
#pragma unroll
        for (int h = 0; h < 3; ++h)
#pragma unroll
          for (int e = 0; e < 4; ++e) m = __builtin_fmaxf(m, __builtin_fabsf(raw[k][h][e]));
      m = ics_wave_max_f32(m);
      if (lane == 0) fscr[wv] = m;
      lds_barrier();     // S0: also orders the previous tile's last gradient phase before the planes are rewritten
#pragma unroll
      for (int w = 0; w < C::NW; ++w) m = __builtin_fmaxf(m, fscr[w]);
      pow2_scale(__builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, m))), s_x, inv_x);
    }
    const float sc = inv_w * inv_x;

    // lane parts of the addresses, re-derived per tile (opaque: nothing of this may be hoisted above the tile loop and kept)
    const int tide = opaque(tid);
    const int eli = tide & 15, elg = (tide >> 4) & 3;
    const int colx = x0 + 16 * wv + eli;
    const int voff = 4 * (8 * elg * pitch + 3 * eli);             // rows t + 8 lg + 2 r of pixel column 16 wv + li
    const int sb = 4 * (y0 * pitch + 3 * (x0 + 16 * wv));
    const int acc_voff = 16 * (tide & 63);
    const int acc_sb = (tile * 4 + wv) * (3 * 2 * 1024);          // accumulator-order image, 32-row layout: [tile][cb][ch][t][lane] float4
    // weights (B operand of the convolution), u planes (A operand), see k_synth_gradk
    const int bo = 8 * elg - eli + 15;
    const bool bzero = bo < 8 || bo > K + 14;
    const uint32_t wsh = bzero ? 0u : (uint32_t)(bo & 1) * 16u;
    const uint32_t wa0 = lds0 + (uint32_t)C::WOFF + 8u * (uint32_t)(bzero ? C::WZERO : ((bo - 8) >> 1));
    const uint32_t conv_a = lds0 + (uint32_t)(C::UOFF + eli * C::ROWB + (16 * wv + 8 * elg) * 2);

    uint32_t fop[2][4];
    auto load_img = [&](int ch) {
      if (ACC) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const u4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_acc, acc_voff, acc_sb + (ch * 2 + t) * 1024, 0);
#pragma unroll
          for (int r = 0; r < 4; ++r) fop[t][r] = v[r];
        }
      } else {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) fop[t][r] = __builtin_amdgcn_raw_buffer_load_b32(rs_f, voff + 4 * ch, sb + 4 * (t + 2 * r) * pitch, 0);
      }
    };

    f4 acc[2];
    // ---- Toeplitz convolution of one channel from plane buffer (ch & 1): fragment q = rows q + 2 i feeds kernel rows q and q - 1 ----
    // One step ahead: the fragment and the raw weight dwords of step q + 1 are requested before the MFMAs of step q (18 registers;
    // the reads are volatile so that they stay there), the funnel shifts follow the MFMAs -- the B slot they fill is read by them.
    auto conv_phase = [&](auto chc) {
      constexpr int ch = decltype(chc)::value;
      constexpr uint32_t PB = (uint32_t)((ch & 1) * 2 * C::PLANE);
      typedef const volatile __attribute__((address_space(3))) u4* lds_vu4p;
      typedef const volatile __attribute__((address_space(3))) u2* lds_vu2p;
      acc[0] = (f4){0.f, 0.f, 0.f, 0.f}; acc[1] = (f4){0.f, 0.f, 0.f, 0.f};
      h8 Bh[2], Bl[2];
      u2 d[5];
      u4 Ah, Al, nAh, nAl;
      auto issue = [&](int q) {     // operands of step q: fragment q, kernel row q
        const uint32_t off = (uint32_t)(C::cls_off(q & 1) + (q >> 1) * C::ROWB);
        nAh = *reinterpret_cast<lds_vu4p>(conv_a + PB + off);
        nAl = *reinterpret_cast<lds_vu4p>(conv_a + PB + C::PLANE + off);
        if (q < K) {
          const lds_vu2p r = reinterpret_cast<lds_vu2p>(wa0 + (uint32_t)((ch * K + q) * 2 * C::WROWB));
#pragma unroll
          for (int k = 0; k < 5; ++k) d[k] = r[k];
        }
      };
      auto finishB = [&](int q) {   // kernel row q enters with fragment q (set 0) and leaves with fragment q + 1 (set 1)
        const u4 wh = {f_align(d[1].x, d[0].x, wsh), f_align(d[2].x, d[1].x, wsh), f_align(d[3].x, d[2].x, wsh), f_align(d[4].x, d[3].x, wsh)};
        const u4 wl = {f_align(d[1].y, d[0].y, wsh), f_align(d[2].y, d[1].y, wsh), f_align(d[3].y, d[2].y, wsh), f_align(d[4].y, d[3].y, wsh)};
        Bh[q & 1] = __builtin_bit_cast(h8, wh);
        Bl[q & 1] = __builtin_bit_cast(h8, wl);
      };
      issue(0);
      finishB(0);
      Ah = nAh; Al = nAl;
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < C::NQ; ++q) {
        if (q + 1 < C::NQ) issue(q + 1);
#pragma unroll
        for (int term = 0; term < 3; ++term)
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const int ka = q - t;
            if (ka < 0 || ka >= K) continue;
            acc[t] = f_mfma32(__builtin_bit_cast(h8, term == 2 ? Al : Ah), term == 1 ? Bl[ka & 1] : Bh[ka & 1], acc[t]);
          }
        if (q + 1 < K) finishB(q + 1);
        Ah = nAh; Al = nAl;
        __builtin_amdgcn_sched_barrier(0);   // one step's operands at a time: registers are scarce at three waves per SIMD
      }
    };

    // ---- e'(ch) = conv - image (0 outside the M x N interior), kept in `acc`; returns the wave's max |e'| ---------------
    auto residual = [&](int ch) -> float {
      float m = 0.f;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int y = y0 + t + 8 * elg + 2 * r;
          const bool in = y < C::PAD + a.g.M && colx < C::PAD + a.g.N;
          const float e = in ? __fsub_rn(acc[t][r] * sc, __uint_as_float(fop[t][r])) : 0.f;
          acc[t][r] = e;
          m = __builtin_fmaxf(m, __builtin_fabsf(e));
          if (store_e && in) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(e), rs_o, voff + 4 * ch, sb + 4 * (t + 2 * r) * pitch, 0);
        }
      return ics_wave_max_f32(m);
    };

    // ---- e' -> fp16 (hi, lo) planes (rows t + 8 lg + 2 r), as in k_synth_gradk ---------------------------------------------
    auto write_e = [&](float s_e) {
      typedef __attribute__((address_space(3))) uint32_t* lds_wp;
      const uint32_t ew = lds0 + (uint32_t)(C::EOFF + 8 * elg * C::EROWB + 8 * ((16 * wv + eli + 8) >> 1) + 4 * (eli & 1));
      const bool odd = (eli & 1) != 0;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float x = acc[t][r] * s_e;
          const _Float16 xh = (_Float16)x;
          const _Float16 xl = (_Float16)(x - (float)xh);
          const uint32_t P = (uint32_t)__builtin_bit_cast(unsigned short, xh) | ((uint32_t)__builtin_bit_cast(unsigned short, xl) << 16);
          const uint32_t Q = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)P, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
          const uint32_t w = odd ? ((Q >> 16) | (P & 0xFFFF0000u)) : ((P & 0xFFFFu) | (Q << 16));
          *reinterpret_cast<lds_wp>(ew + (uint32_t)((t + 2 * r) * C::EROWB)) = w;
        }
    };

    // ---- PSF gradient of one channel: the 8 residual rows of this wave as 4 pairs sharing one u fragment (see k_synth_gradk).
    // 24 steps (row, column chunk), three MFMAs each; the e' dwords of step s + 1 are requested before the MFMAs of step s, and every
    // u fragment is re-requested into its own registers right after its last use (two steps before the next pair needs it): no
    // second register set.  The other two waves of the SIMD fill what is left of the gaps.
    auto gradk_phase = [&](auto chc, float scale) {
      constexpr int ch = decltype(chc)::value;
      constexpr uint32_t PB = (uint32_t)((ch & 1) * 2 * C::PLANE);
      typedef const volatile __attribute__((address_space(3))) u4* lds_vu4p;
      typedef const volatile __attribute__((address_space(3))) u2* lds_vu2p;
      // lane row m reads u row (y + 1) + 2 pad - m, y = 8 wv + 2 p: its class is a lane constant, pairs advance one row inside it
      const int mm = eli < 2 * C::PAD + 1 ? eli : 2 * C::PAD + 1;
      const int r0 = 8 * wv + 1 + 2 * C::PAD - mm;
      const uint32_t ga = lds0 + (uint32_t)(C::UOFF + ((r0 & 1) ? C::OFF1 : 0) + (r0 >> 1) * C::ROWB + 16 * elg) + PB;
      const int tb = eli < K ? eli : K - 1;
      uint32_t gb[3], gsh[3];
#pragma unroll
      for (int X = 0; X < 3; ++X) {
        int sx = X < 2 ? 32 * X + 8 * elg + tb - 2 * C::PAD : 64 + 4 * elg + tb - 2 * C::PAD;
        sx = sx <= -8 ? -8 : (sx >= 64 ? 64 : sx);
        gsh[X] = (uint32_t)(sx & 1) * 16u;
        gb[X] = lds0 + (uint32_t)(C::EOFF + 8 * wv * C::EROWB + 8 * ((sx + 8) >> 1));
      }
      f4 g[2][2];
#pragma unroll
      for (int h = 0; h < 2; ++h) { g[h][0] = (f4){0.f, 0.f, 0.f, 0.f}; g[h][1] = (f4){0.f, 0.f, 0.f, 0.f}; }
      u4 Ah[2], Al[2];
      u2 A2h, A2l;
      u2 d[5];
      auto issueA = [&](int pair, int X) {
        const uint32_t ar = ga + (uint32_t)(pair * C::ROWB);
        if (X < 2) { Ah[X] = *reinterpret_cast<lds_vu4p>(ar + 64 * X); Al[X] = *reinterpret_cast<lds_vu4p>(ar + C::PLANE + 64 * X); }
        else { A2h = *reinterpret_cast<lds_vu2p>(ar + 128 - 8 * elg); A2l = *reinterpret_cast<lds_vu2p>(ar + C::PLANE + 128 - 8 * elg); }
      };
      auto issueB = [&](int st) {
        const int i = st / 3, X = st - 3 * i;
        const lds_vu2p ep = reinterpret_cast<lds_vu2p>(gb[X] + (uint32_t)(i * C::EROWB));
#pragma unroll
        for (int k = 0; k < 5; ++k) if (X < 2 || k < 3) d[k] = ep[k];
      };
      issueA(0, 0); issueA(0, 1); issueA(0, 2);
      issueB(0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int st = 0; st < 24; ++st) {
        const int i = st / 3, X = st - 3 * i, hp = i & 1;
        h8 Bh, Bl;
        h4 B2h, B2l;
        if (X < 2) {
          const u4 wh = {f_align(d[1].x, d[0].x, gsh[X]), f_align(d[2].x, d[1].x, gsh[X]), f_align(d[3].x, d[2].x, gsh[X]), f_align(d[4].x, d[3].x, gsh[X])};
          const u4 wl = {f_align(d[1].y, d[0].y, gsh[X]), f_align(d[2].y, d[1].y, gsh[X]), f_align(d[3].y, d[2].y, gsh[X]), f_align(d[4].y, d[3].y, gsh[X])};
          Bh = __builtin_bit_cast(h8, wh); Bl = __builtin_bit_cast(h8, wl);
        } else {
          const u2 wh = {f_align(d[1].x, d[0].x, gsh[2]), f_align(d[2].x, d[1].x, gsh[2])};
          const u2 wl = {f_align(d[1].y, d[0].y, gsh[2]), f_align(d[2].y, d[1].y, gsh[2])};
          B2h = __builtin_bit_cast(h4, wh); B2l = __builtin_bit_cast(h4, wl);
        }
        if (st + 1 < 24) issueB(st + 1);                     // (d is free: the shifts above consumed it)
        if (X < 2) {
          g[hp][0] = f_mfma32(__builtin_bit_cast(h8, Ah[X]), Bh, g[hp][0]);
          g[hp][1] = f_mfma32(__builtin_bit_cast(h8, Ah[X]), Bl, g[hp][1]);
          g[hp][0] = f_mfma32(__builtin_bit_cast(h8, Al[X]), Bh, g[hp][0]);
        } else {
          g[hp][1] = f_mfma16(__builtin_bit_cast(h4, A2h), B2h, g[hp][1]);
          g[hp][0] = f_mfma16(__builtin_bit_cast(h4, A2h), B2l, g[hp][0]);
          g[hp][1] = f_mfma16(__builtin_bit_cast(h4, A2l), B2h, g[hp][1]);
        }
        // the fragment of chunk X was last used by the odd row of its pair: the next pair's goes into the same registers now
        if (hp == 1 && i + 1 < 8) issueA((i + 1) >> 1, X);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float s1 = g[1][0][r] + g[1][1][r];
        const float s0n = r < 3 ? g[0][0][r + 1] + g[0][1][r + 1] : 0.f;
        tot[ch][r] += (s1 + s0n) * scale;
      }
      carry[ch] += (g[0][0][0] + g[0][1][0]) * scale;
    };

    // ================================ the tile ======================================================================
    unsigned char* const up = lds + C::UOFF;
    convert_channel2<C, 0>(raw, s_x, up, opaque(tid));
    lds_barrier();                                                     // planes of channel 0 visible
    load_img(0);
    conv_phase(std::integral_constant<int, 0>{});
    float s_e, inv_e, me;

#define ICS_FUSED2_CHANNEL(CH)                                                                                          \
    me = residual(CH);                                                                                                  \
    if (lane == 0) fscr[8 + 4 * (CH) + wv] = me;                                                                        \
    lds_barrier();     /* tile maximum; every wave is past the previous gradient phase: e' planes and u buffer free */  \
    me = __builtin_fmaxf(__builtin_fmaxf(fscr[8 + 4 * (CH)], fscr[9 + 4 * (CH)]), __builtin_fmaxf(fscr[10 + 4 * (CH)], fscr[11 + 4 * (CH)])); \
    pow2_scale(__builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, me))), s_e, inv_e);      \
    write_e(s_e);

    ICS_FUSED2_CHANNEL(0)
    convert_channel2<C, 1>(raw, s_x, up + 2 * C::PLANE, opaque(tid));
    lds_barrier();                                                     // e'(0) and planes(1) visible
    gradk_phase(std::integral_constant<int, 0>{}, inv_x * inv_e);
    load_img(1);
    conv_phase(std::integral_constant<int, 1>{});

    ICS_FUSED2_CHANNEL(1)
    convert_channel2<C, 2>(raw, s_x, up, opaque(tid));
    lds_barrier();                                                     // e'(1) and planes(2) visible
    gradk_phase(std::integral_constant<int, 1>{}, inv_x * inv_e);
    load_img(2);
    auto prefetch = [&]() {
      if (tile + nx < band1) {
        const int nt = tile + nx;
        const int nyi = nt / tpr, nxi = nt - nyi * tpr;
        load_raw2<C>(raw, rs_in, 4 * ((a.g.ay + TORG + nyi * C::TH - C::PAD) * pitch + 3 * (a.g.ax + TORG + nxi * C::TW - C::PAD)), opaque(tid), pitch);
      }
    };
    if (ICS_FUSED2_PREFETCH == 2) prefetch();                          // the rows of the next tile: in flight during conv(2), gradk(2)
    conv_phase(std::integral_constant<int, 2>{});

    ICS_FUSED2_CHANNEL(2)
    lds_barrier();                                                     // e'(2) visible
    if (ICS_FUSED2_PREFETCH == 1) prefetch();                          // ... during gradk(2) only
    gradk_phase(std::integral_constant<int, 2>{}, inv_x * inv_e);
    if (ICS_FUSED2_PREFETCH == 0) prefetch();                          // ... not at all: the other two workgroups of the CU cover the latency
#undef ICS_FUSED2_CHANNEL
  }

  // ---- cross-wave reduction (fixed order) and partial write, one channel per pass: as k_synth_gradk -------------------------
  float* red = reinterpret_cast<float*>(lds);
  float* dst = a.partial + (size_t)blockIdx.x * (3 * 16 * 16);
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wv * 256 + r * 64 + lane] = tot[c][r];
    red[C::NW * 256 + wv * 64 + lane] = carry[c];
    __syncthreads();
    {
      const int v = tid;
      float s = red[v];
#pragma unroll
      for (int w = 1; w < C::NW; ++w) s += red[w * 256 + v];
      const int l = v & 63, r = (v >> 6) & 3;
      if (r == 3 && l < 48) {
#pragma unroll
        for (int w = 0; w < C::NW; ++w) s += red[C::NW * 256 + w * 64 + l + 16];
      }
      const int ta = 4 * (l >> 4) + r, tb = l & 15;
      dst[(c * 16 + ta) * 16 + tb] = s;
    }
  }

}

template <int K>
hipError_t launch_k(const IcsFusedArgs& a, int nblocks, hipStream_t s) {
  const int dev = ics_current_device();
  const bool ACC = a.facc != nullptr;
  if (a.rs == 2) {   // 32-row tiles, three workgroups per CU
    using C = FCfg2<K>;
    static std::atomic<bool> configured[2][ICS_MAX_DEVICES];
    auto kern = ACC ? k_synth_gradk2<K, true> : k_synth_gradk2<K, false>;
    if (hipError_t e = ics_configure_lds(configured[ACC ? 1 : 0], dev, kern, C::LDS_BYTES); e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(nblocks), dim3(C::NT), C::LDS_BYTES, s, a);
    return hipGetLastError();
  }
  using C = FCfg<K>;
  static std::atomic<bool> configured[2][ICS_MAX_DEVICES];
  auto kern = ACC ? k_synth_gradk<K, true> : k_synth_gradk<K, false>;
  if (hipError_t e = ics_configure_lds(configured[ACC ? 1 : 0], dev, kern, C::LDS_BYTES); e != hipSuccess) return e;
  // every workgroup of the grid writes its partial block (the reduction reads `nblocks` of them): workgroups without a tile
  // write zeros
  hipLaunchKernelGGL(kern, dim3(nblocks), dim3(C::NT), C::LDS_BYTES, s, a);
  return hipGetLastError();
}

}  // namespace

bool ics_synth_gradk_supported(int K) { return K >= 3 && K <= 15 && (K & 1); }

hipError_t ics_launch_synth_gradk(const IcsFusedArgs& a, int nblocks, hipStream_t s) {
  if (!a.bt) return hipErrorInvalidValue;
  switch (a.g.K) {
    case 3: return launch_k<3>(a, nblocks, s);
    case 5: return launch_k<5>(a, nblocks, s);
    case 7: return launch_k<7>(a, nblocks, s);
    case 9: return launch_k<9>(a, nblocks, s);
    case 11: return launch_k<11>(a, nblocks, s);
    case 13: return launch_k<13>(a, nblocks, s);
    case 15: return launch_k<15>(a, nblocks, s);
    default: return hipErrorInvalidValue;
  }
}
