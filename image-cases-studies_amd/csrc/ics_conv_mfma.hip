// ics_conv_mfma.hip -- the two PSF convolutions of one Richardson-Lucy inner iteration on the gfx950
// matrix cores (PSF sizes 3..15).  Same contract as ics_conv.hip (modes 0 and 1):
//
//   mode 0 (A1+A2, lib/deconvolution.pyx:477-488):  error = convolve(u, psf, "valid") - image
//   mode 1 (A3,    lib/deconvolution.pyx:490-491):  gradu = convolve(error, rot180(psf), "full")
//           + the reductions of A7 (pyx:523-524)
//
//     out[y, x, c] = sum_{a,b<K} W[a, b, c] * in[y + a - pad, x + b - pad, c]
//
// Why a second kernel: the packed-fp32 VALU kernel of ics_conv.hip tops out at ~88 TFLOP/s (of the 135
// that v_pk_fma_f32 can issue) and is the largest item of the iteration.  The matrix cores have no fast
// fp32 mode on CDNA4 (v_mfma_f32_16x16x4_f32 = 157 TF), but v_mfma_f32_16x16x32_f16 sustains 1.9 PF
// with fp32 accumulation.  Every fp32 operand is therefore split into two fp16 terms,
//     x * s = hi + lo,  hi = fp16(x*s),  lo = fp16(x*s - hi)          (s = a power of two per tile / per PSF)
// which keeps 22 significand bits, and the product is evaluated as hi*hi + hi*lo + lo*hi (three MFMAs,
// the lo*lo term is below 2^-22).  Measured against float64 the result is as close as the sequential fp32
// FMA chain of the VALU kernel (oracle comparison in tests/test_gpu_stages.py, same tolerance for both).
//
// Formulation.  Along x the convolution is a banded Toeplitz product, one per kernel row a and channel c:
//     out_a[y, 16cb + j] = sum_{k<32} in[y + a - pad, 16cb - pad + k] * B_a[k][j],   B_a[k][j] = W[a][k - j][c]
// (zero outside 0 <= k-j < K; 16 + K - 1 <= 32 input columns per 16 output columns, hence K <= 17; built for K <= 15, where the weight rows still fit in LDS beside the planes).
// The MFMA's M dimension runs over 16 image rows, and the rows of one fragment are taken 4 apart:
//     fragment q, lane row i  <->  input row q + 4i (tile-relative)       q = 0 .. K+2
// so that the product with B_a lands on output rows (q - a) + 4i: fragment q feeds the four accumulator
// sets t = q - a = 0..3 IN PLACE, i.e. one LDS fragment read serves up to 4 kernel rows x 3 split terms =
// 12 MFMAs (a row-contiguous fragment would serve 3 and the kernel would be LDS-bound).
//
// Kernel shape: persistent 4-wave workgroups walking 64x64-pixel tiles, two per CU -- or, with fragment rows 2 apart (two
// accumulator sets per wave instead of four), 32x64 tiles, three per CU: launch_k() below picks per PSF size and frame.
//   * LDS (75 KB): the tile + halo as six fp16 planes (channel x hi/lo), rows grouped by y mod 4 so that the
//     16 lane rows of a fragment are consecutive 160-B LDS rows (conflict-free for the b128 lane groups of
//     gfx950, MI355X_MICROARCH.md LDS).  Two workgroups per CU: one's memory phases (conversion, epilogue)
//     overlap the other's matrix phase.
//   * wave w owns the 16-column block w of the tile for all three channels (12 accumulators).
//   * Toeplitz fragments are never stored: the weights sit in LDS as zero-padded rows (2 x 64 B per (c, a): the hi and lo
//     split terms interleaved dword by dword, 5.6 KB at K = 15); a lane's 8 consecutive halves start at half 8g - j + 15 of
//     the row: it reads the five (hi, lo) dword pairs that contain them (ds_read_b64) and funnel-shifts by the parity.
//     (Full 1-KiB fragments from global memory made the kernel L1-bound; a ds_bpermute gather from a row image cost ~5 LDS
//     cycles per bpermute; separate hi and lo rows cost 12 LDS instructions per step instead of 7.)
//   * issue order inside a step (12 MFMAs): the LDS reads of the next step go into the shadows of the first MFMAs, the funnel
//     shifts into the shadows of the last ones (sched_group_barrier); the reads are volatile so that they stay where they are
//     requested -- as plain loads they were sunk to the shifts and every step waited out the LDS latency.
//   * the fp32 HWC rows of the NEXT tile are requested into registers (2 waves per SIMD -> 256 VGPRs) before
//     the MFMA loop, which itself issues no vector-memory load (they return in order): HBM latency is covered
//     by the matrix phase; conversion to the fp16 planes happens after the epilogue of the current tile.
//   * epilogue straight from the accumulators: a lane holds one 12-byte HWC pixel per accumulator row, operands arrive and
//     results leave as dwordx3 (16 lanes = 192 contiguous bytes), same arithmetic as the VALU kernel (residual /
//     back-projection + step-size reductions).  Two barriers per tile (scale, planes written).
#include "ics_common.h"
#include "ics_image_acc.h"
#include <stdlib.h>
#include <type_traits>

#ifndef ICS_MFMA_INTERLEAVE
#define ICS_MFMA_INTERLEAVE 1
#endif
#ifndef ICS_EPI_LOAD_AUX
#define ICS_EPI_LOAD_AUX 0    /* cache policy of the epilogue operand loads (2 = nt measured slower: the update pass that follows finds less of u / ut in the memory-side cache) */
#endif
#ifndef ICS_EPI_LOAD_AUX0
#define ICS_EPI_LOAD_AUX0 ICS_EPI_LOAD_AUX   /* the same for mode 0 (image operand) */
#endif
#ifndef ICS_RAW_AUX
#define ICS_RAW_AUX 0         /* cache policy of the tile loads */
#endif
#ifndef ICS_EPI_STORE_AUX
#define ICS_EPI_STORE_AUX 0
#endif
#ifndef ICS_EPI_TB
#define ICS_EPI_TB(mode) ((mode) == 0 ? 4 : 2)   /* mode 1 carries two operand frames: two batches keep it spill-free */
#endif
#ifndef ICS_EPI_EARLY1
#define ICS_EPI_EARLY1 1      /* 16-row tiles (small frames, one tile per workgroup), mode 0: the image operand of the residual is requested with the tile's rows, at the top of the kernel (-0.2 us of 6.9).  The same for mode 1's u / majoriser operands measured +0.7 ... 1.5 us -- more requests in front of the rows the kernel waits for -- and is not built */
#endif
#ifndef ICS_EPI_EARLY
#define ICS_EPI_EARLY 1       /* mode 0, 32-row tiles, accumulator-order image: request the image operand BEFORE the matrix phase (130 of 168 VGPRs in use: room for its 24) */
#endif
#ifndef ICS_MFMA_PAIRS
#define ICS_MFMA_PAIRS 2   /* two-window sizes up to 33 x 33 (8-wave kernels): two kernel rows per three MFMA windows, see MCfg::PAIR.
                              1: the two waves of a column block split the accumulator sets (t = half, half + 2), all kernel rows each;
                              2: they split the row pairs (items) and keep all four sets: a B fragment then serves four sets */
#endif
#ifndef ICS_MFMA_WSPLIT
#define ICS_MFMA_WSPLIT 1   /* see MCfg::WSPLIT */
#endif
#ifndef ICS_MFMA_NO_RS1
#define ICS_MFMA_NO_RS1 0
#endif
#ifndef ICS_MFMA_ALL_RS
#define ICS_MFMA_ALL_RS 0  /* tools/: build both tile heights for every PSF size (ICS_TEST_CONV_RS=2|4 then picks one) */
#endif
#ifndef ICS_MFMA_ABLATE
#define ICS_MFMA_ABLATE 0  /* tools/bench_conv_mfma.hip: 1 = no MFMA loop, 2 = no conversion, 4 = no epilogue, 64 = two of the three split terms only */
#endif

// phase timing probe (tools/bench_conv_mfma.hip -DICS_MFMA_TIMING): per-wave cycle totals between the marks
#ifdef ICS_MFMA_TIMING
__device__ unsigned long long ics_mfma_ticks[11];
#define ICS_TICK_INIT unsigned long long tk_prev = __builtin_readcyclecounter(), tk_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define ICS_TICK(i) do { const unsigned long long tk_now = __builtin_readcyclecounter(); tk_acc[i] += tk_now - tk_prev; tk_prev = tk_now; } while (0)
#define ICS_TICK_FLUSH do { if ((threadIdx.x & 63) == 0) { for (int i = 0; i < 10; ++i) atomicAdd(&ics_mfma_ticks[i], tk_acc[i]); atomicAdd(&ics_mfma_ticks[10], 1ull); } } while (0)
#elif defined(ICS_MFMA_TRACE)
// phase timeline (tools/bench_conv_mfma.hip -DICS_MFMA_TRACE): lane 0 of every wave records (100 MHz wall clock << 8 | mark)
// at each mark; entry 0 = HW_ID | XCC_ID << 32.  1024 entries per wave.
__device__ unsigned long long* ics_trace_buf;
#define ICS_TICK_INIT unsigned long long* tr_ = ics_trace_buf + ((size_t)blockIdx.x * 4 + wv) * 1024; int tri_ = 0; \
  if (lane == 0) { tr_[tri_++] = (unsigned long long)__builtin_amdgcn_s_getreg(63492) | ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32); tr_[tri_++] = (wall_clock64() << 8) | 15; }
#define ICS_TICK(i) do { if (lane == 0 && tri_ < 1023) tr_[tri_++] = (wall_clock64() << 8) | (i); } while (0)
#define ICS_TICK_FLUSH do { if (lane == 0) tr_[tri_] = 0; } while (0)
#else
#define ICS_TICK_INIT
#define ICS_TICK(i)
#define ICS_TICK_FLUSH
#endif

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
typedef uint32_t u3 __attribute__((ext_vector_type(3)));

// RS = row stride of a fragment = accumulator sets per wave and channel: a tile is 16 * RS rows high.
//   RS = 4 (64 x 64 tiles, two workgroups per CU up to K = 15): one A-fragment read feeds 12 MFMAs per window.
//   RS = 2 (32 x 64 tiles): half the planes, a third of the registers less -- three workgroups per CU up to K = 17 and two
//   instead of one at K = 19, 21; twice the LDS reads per MFMA and 9 % more staged bytes.  launch_k() picks (measured).
// NH = 2 (K >= 23, where the planes leave room for one workgroup per CU only): 8 waves per workgroup; the two waves of a column
//   block split the kernel rows ([0, K/2] and the rest), exchange their partial sums through LDS after the matrix phase and run
//   the epilogue of two accumulator sets each.  Two waves per SIMD instead of one: every phase of a wave (conversion, LDS
//   latency of the fragment reads, epilogue) finds another wave to overlap with.
template <int K, int RS_, int NH_ = 1>
struct MCfg {
  static constexpr int PAD = K / 2;
  static constexpr int RS = RS_, NH = NH_;
  static constexpr int KSPLIT = NH == 2 ? (K + 1) / 2 : K;   // first kernel row of the second half
  static constexpr int TH = 16 * RS, TW = 64, NCB = TW / 16;
  static constexpr int NW = 4 * NH, NT = 64 * NW;
  static constexpr int LROWS = TH + K - 1;       // input rows of the tile
  // LDS rows are grouped by (y mod 4): class c holds rows c, c+4, ... contiguously, classes back to back
  static constexpr int cls_rows(int c) { return (LROWS - c + RS - 1) / RS; }
  static constexpr int cls_base(int c) { return c == 0 ? 0 : cls_base(c - 1) + cls_rows(c - 1); }
  // 16 output columns need 16 + K - 1 input columns: one 32-wide MFMA window (NCH = 1, K <= 17) or two (K <= 49)
  static constexpr int NCH = (16 + K - 1 <= 32) ? 1 : 2;
  static constexpr int LCOLS = TW + 32 * NCH - 16;   // staged columns: [x0 - PAD, x0 - PAD + 80 / 112)
  static constexpr int ROWB = 2 * LCOLS;             // 160 / 224 bytes per LDS row: conflict-free for the fragment reads
  static constexpr int PLANE = LROWS * ROWB;     // bytes per (channel, hi/lo) plane
  static constexpr int DATA = 6 * PLANE;
  static constexpr int SCRATCH = DATA;
  // weight rows in LDS: halves 8 .. K+24 of the zero-padded row Wp[idx] = W[idx - 15] (the taps sit at local
  // halves 7 .. K+6, at least ten zeros follow): every 8-half window that meets a tap lies inside, and the
  // all-zero windows are redirected to the zero tail
  static constexpr int WROWB = (2 * (K + 17) + 3) & ~3;
  static constexpr int WZERO = (K + 7) / 2;      // first all-zero dword of a row
  static constexpr int WLDS = 3 * K * 2 * WROWB; // = the global weight table built by k_psf (ics_common.h), copied verbatim
  // PAIR_ITEMS, where the LDS has room (K <= 31): the weight rows are kept as FOUR plain rows per (channel, kernel row) instead of
  // one hi / lo dword-interleaved row -- hi and lo, each once as it is and once moved up by one half.  A lane then finds its 8
  // consecutive halves dword-aligned in the copy of its parity and reads them with two 4-byte-aligned 8-byte loads per split term
  // (ds_read2_b32) straight into the MFMA operand registers: 4 LDS instructions and no funnel shift per B fragment instead of
  // 5 + 8 (the pair loops run at the wave's issue limit: 2.5 other instructions per MFMA, profiles/r04_6144_31_mfma_counters.json).
  static constexpr bool WSPLIT_WANTED = ICS_MFMA_PAIRS == 2 && ICS_MFMA_WSPLIT && NH_ == 2 && RS_ == 4 && (16 + K - 1 > 32) && (16 + K - 1 <= 48);
  static constexpr bool WSPLIT = WSPLIT_WANTED && (size_t)SCRATCH + 256 + 2 * WLDS <= 160 * 1024;
  static constexpr int WLDS_USED = WSPLIT ? 2 * WLDS : WLDS;
  static constexpr size_t LDS_BYTES = SCRATCH + 256 + WLDS_USED;
  static constexpr int WGS_CAP = RS == 4 ? 2 : 3;                      // register budget: 256 / 168 VGPRs
  static constexpr int WGS = (160 * 1024 / LDS_BYTES) < WGS_CAP ? (160 * 1024 / LDS_BYTES) : WGS_CAP;   // workgroups (of 4 waves) per CU
  // Row pairs (round 4).  With two windows a kernel row costs 2 x 32 columns of MFMA depth for its 16 + K - 1 <= 48 input columns,
  // and the matrix loop of these kernels runs at 85 % of the MFMA issue rate (without it 0.35 of 1.07 ms at 6144^2 / 31 x 31, with
  // two of the three split terms 0.83: tools/bench_conv_mfma.hip -DICS_MFMA_ABLATE=64).  Two consecutive kernel rows a, a + 1 of one
  // accumulator set read input rows r and r + 1; their 2 x 48 columns fill THREE 32-deep windows exactly:
  //     window 0 = row r cols [0, 32) | window m = row r cols [32, 48) + row r + 1 cols [0, 16) | window 2 = row r + 1 cols [16, 48)
  // -- 24 % fewer MFMAs at K = 31.  The mixed window's A fragment takes lane groups 0, 1 from one LDS row class and 2, 3 from the next
  // (a per-lane base address), its B fragment likewise from the weight rows of a and a + 1.  A fragment pair (q, q + 1) then serves
  // the accumulator sets t = q - a with a EVEN only, so the two waves of a column block no longer split the kernel rows but the
  // sets: wave `half` owns t = half and half + 2 for all rows -- no exchange of partial sums, two workgroup barriers less per tile.
  static constexpr bool PAIR = ICS_MFMA_PAIRS && NH == 2 && RS == 4 && (16 + K - 1 > 32) && (16 + K - 1 <= 48);
  static constexpr bool PAIR_SETS = PAIR && ICS_MFMA_PAIRS == 1;   // waves own accumulator sets (no exchange of partial sums)
  static constexpr bool PAIR_ITEMS = PAIR && ICS_MFMA_PAIRS == 2;  // waves own row pairs, all four sets (partial sums exchanged as in the classic split)
  static constexpr int NQ = K + RS - 1;          // fragments per (channel, column block)
  static constexpr int XG = LCOLS / 4;           // 4-pixel groups per staged row
  static constexpr int NTASK = LROWS * XG;
  static constexpr int NIT = (NTASK + NT - 1) / NT;
  static constexpr int ETASK = TH * (TW / 4);    // epilogue tasks: one row x 4 pixels
  static constexpr int EIT = (ETASK + NT - 1) / NT;
  static_assert(16 + K - 1 <= 32 * NCH, "MFMA windows cover the taps of 16 output columns");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) { return ics_wave_max_u32(v); }

// power-of-two scale that brings a maximum magnitude m into [2^14, 2^15) (fp16 overflows at 65504);
// 1 for m = 0 / Inf / NaN.  `inv` is the exact inverse.
__device__ __forceinline__ void pow2_scale(float m, float& s, float& inv) {
  const uint32_t e = (__float_as_uint(m) >> 23) & 0xFFu;
  uint32_t sb = 127u;
  if (m > 0.f && e != 255u) { sb = 268u - e; sb = sb > 240u ? 240u : sb; }
  s = __uint_as_float(sb << 23);
  inv = __uint_as_float((254u - sb) << 23);
}

// Buffer addressing (SGPR resource + 32-bit lane offset + SGPR/immediate offset): with flat 64-bit pointers
// the compiler materialised one 64-bit VGPR base per load and spilled them.
#define ICS_BUF_WORD3 0x00020000  /* gfx9 raw buffer: DATA_FORMAT = 32 */
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7FFFFFFF, ICS_BUF_WORD3);
}

// the staged rows of a tile: task t = (row, 4-pixel group) -> three dwordx4 loads (4-byte aligned).
// `soff` = wave-uniform byte offset of the tile's first staged element.
template <typename C>
__device__ __forceinline__ void load_raw(f32x4u (&v)[C::NIT][3], __amdgpu_buffer_rsrc_t rs, int soff, int tid, int pitch) {
#pragma unroll
  for (int k = 0; k < C::NIT; ++k) {
    int t = tid + k * C::NT;
    t = t < C::NTASK ? t : C::NTASK - 1;  // clamp instead of predicating the load
    const int row = t / C::XG, xg = t - row * C::XG;
    const int toff = 4 * (row * pitch + 12 * xg);
#pragma unroll
    for (int h = 0; h < 3; ++h) v[k][h] = __builtin_bit_cast(f32x4u, __builtin_amdgcn_raw_buffer_load_b128(rs, toff + 16 * h, soff, ICS_RAW_AUX));
  }
}

// workgroup barrier that waits for this wave's LDS traffic only (__syncthreads() also waits for the global loads in flight)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// a copy of `x` the optimiser cannot trace back: values derived from it are recomputed where they are used
// instead of being hoisted out of the tile loop (where they were spilled -- and a scratch reload waits on
// vmcnt, i.e. on the whole prefetch in flight)
__device__ __forceinline__ int opaque(int x) { asm volatile("" : "+v"(x)); return x; }

template <int K, int MODE, int RS, int NH>
__global__ __launch_bounds__(256 * NH) __attribute__((amdgpu_waves_per_eu(MCfg<K, RS, NH>::WGS * NH, MCfg<K, RS, NH>::WGS * NH))) void k_conv_mfma(IcsConvArgs a) {
  using C = MCfg<K, RS, NH>;
  static_assert(NH == 1 || MCfg<K, RS, NH>::WGS == 1, "the row split is for the one-workgroup-per-CU sizes (RS / 2 accumulator sets per wave in the epilogue)");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  float* fscr = reinterpret_cast<float*>(lds + C::SCRATCH);
  const int tid = threadIdx.x;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cb = wv & 3, half = wv >> 2;                      // column block of this wave; which kernel rows it takes (NH = 2)
  const int lane = tid & 63, li = lane & 15, lg = lane >> 4;
  const int pitch = a.g.pitch;
  ICS_TICK_INIT;

  // persistent tile walk: workgroup b runs on XCD b % 8 (observed dispatch); every XCD owns one contiguous
  // band of tiles so that the halos shared by neighbouring tiles hit in that XCD's L2
  // Tile grid.  Mode 1 writes the whole u-frame: tiles start at its origin.  Mode 0 writes the M x N interior only: its
  // tiles start at (PAD, PAD), so that a 4096^2 image is 64 x 64 tiles -- 8 per persistent workgroup -- instead of the 65 x 65
  // (8.25 per workgroup = 9 rounds, the last column and row nearly empty) of the u-frame grid.
  constexpr int TORG = MODE == 0 ? C::PAD : 0;
  const int tpr = MODE == 0 ? (a.g.N + C::TW - 1) / C::TW : a.g.tiles_x;   // tiles per row
  const int ntiles = tpr * (MODE == 0 ? (a.g.M + C::TH - 1) / C::TH : (a.g.uM + C::TH - 1) / C::TH);
  const int xend = MODE == 0 ? C::PAD + a.g.N : a.g.uN;                    // first column without output
  const int nb = (int)gridDim.x < 8 ? (int)gridDim.x : 8;         // bands (= XCDs when the grid covers them all)
  const int xcd = blockIdx.x % nb, kx = blockIdx.x / nb;
  const int nx = ((int)gridDim.x + nb - 1 - xcd) / nb;            // workgroups walking this band
  const int band0 = ics_band_begin(ntiles, (int)gridDim.x, nb, xcd), band1 = ics_band_begin(ntiles, (int)gridDim.x, nb, xcd + 1);
  // Tile walk inside a band.  Static: workgroup kx takes tiles band0 + kx, + nx, ...  Dynamic (a.sched): the first tile is the
  // static one, every further tile is claimed from the band's counter when the current one starts (early enough for the
  // register prefetch) -- the back-projection's 65 x 65 tiles of a 4096^2 frame are 8.25 per workgroup and the edge tiles
  // cost a quarter of an interior one: statically a quarter of the workgroups ran a ninth round while the rest idled.
  int tile = band0 + kx;
  int* lds_next = reinterpret_cast<int*>(fscr + 8);   // two slots, by tile parity: a fast wave may claim for tile i + 1 before a slow one has read the claim of tile i
  int parity = 0;

  // The first tile's rows are requested before anything else: on small frames (one tile per workgroup: the 255-px windows of
  // deblur_module's blind phase, 512^2) the kernel is a chain of dependent round trips, and the weight rows' trip to the LDS below
  // used to sit in front of this one.
  // frame allocation start = origin - (ay rows + ax pixels); tile offsets are then non-negative
  const __amdgpu_buffer_rsrc_t rs_in = make_rsrc(a.in - ((ptrdiff_t)a.g.ay * pitch + 3 * a.g.ax));
  // (and the weight rows ahead of those: loads return in order, so the rows' trip to the LDS below waits for nothing but itself while the
  //  tile rows are still in flight -- as a load / store loop behind them it was a second round trip in series: 2.7 us of the 8.5 us a
  //  255^2 synthesis takes were spent before the first conversion)
  constexpr int WPT = C::WSPLIT ? 1 : (C::WLDS / 4 + C::NT - 1) / C::NT;
  uint32_t wreg[WPT];
  if constexpr (!C::WSPLIT) {
    const uint32_t* tab = reinterpret_cast<const uint32_t*>(a.bt);
#pragma unroll
    for (int k = 0; k < WPT; ++k) { const int i = tid + k * C::NT; wreg[k] = i < C::WLDS / 4 ? tab[i] : 0u; }
  }
  f32x4u raw[C::NIT][3];
  if (tile < band1) {
    const int tyi = tile / tpr, txi = tile - tyi * tpr;
    load_raw<C>(raw, rs_in, 4 * ((a.g.ay + TORG + tyi * C::TH - C::PAD) * pitch + 3 * (a.g.ax + TORG + txi * C::TW - C::PAD)), tid, pitch);
  }
  // 16-row tiles are what frames with fewer tiles than compute units get: every workgroup runs ONE tile and the kernel's time is its chain
  // of dependent round trips, so the synthesis' image operand travels with the rows instead of behind the matrix phase (on frames that fill
  // the device the same move measured nothing: other workgroups cover the latency there; ICS_EPI_EARLY1 above for the back-projection).
  constexpr bool EARLY1 = ICS_EPI_EARLY1 != 0 && RS == 1 && NH == 1 && MODE == 0;
  const __amdgpu_buffer_rsrc_t rs_f = make_rsrc(MODE == 0 ? a.f : a.u);
  const __amdgpu_buffer_rsrc_t rs_t = make_rsrc(MODE == 0 ? a.f : a.ut);
  u3 pre1[4];
  auto request1 = [&](int tl) {
    const int tyi = tl / tpr, txi = tl - tyi * tpr;
    const int x0 = TORG + txi * C::TW, y0 = TORG + tyi * C::TH;
    if (x0 + 16 * cb < xend) {
      const int tide = opaque(tid);
      const int voff = 4 * (4 * C::RS * ((tide >> 4) & 3) * pitch + 3 * (tide & 15)), sb = 4 * (y0 * pitch + 3 * (x0 + 16 * cb));
#pragma unroll
      for (int r = 0; r < 4; ++r) pre1[r] = __builtin_amdgcn_raw_buffer_load_b96(rs_f, voff, sb + 4 * C::RS * r * pitch, ICS_EPI_LOAD_AUX0);
    }
  };
  bool first1 = true;
  if (EARLY1 && tile < band1) request1(tile);
  const float inv_w = *reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(a.bt) + C::WLDS);

  // weight rows -> LDS once per workgroup (the global table is the LDS image)
  {
    uint32_t* ldsW = reinterpret_cast<uint32_t*>(lds + C::SCRATCH + 256);
    const uint32_t* tab = reinterpret_cast<const uint32_t*>(a.bt);
    if constexpr (C::WSPLIT) {
      // [c][a] blocks of four rows of RD dwords: hi, lo, hi moved up one half, lo moved up one half (source: hi dword d at 2d, lo at 2d + 1)
      constexpr int RD = C::WROWB / 4;
      for (int i = tid; i < 3 * K * RD; i += C::NT) {
        const int ca = i / RD, d = i - ca * RD;
        const uint32_t* src = tab + ca * 2 * RD;
        const uint32_t h0 = src[2 * d], l0 = src[2 * d + 1];
        const uint32_t h1 = d + 1 < RD ? src[2 * d + 2] : 0u, l1 = d + 1 < RD ? src[2 * d + 3] : 0u;
        uint32_t* dst = ldsW + ca * 4 * RD + d;
        dst[0] = h0; dst[RD] = l0;
        dst[2 * RD] = __builtin_amdgcn_alignbit(h1, h0, 16); dst[3 * RD] = __builtin_amdgcn_alignbit(l1, l0, 16);
      }
    } else {
#pragma unroll
      for (int k = 0; k < WPT; ++k) { const int i = tid + k * C::NT; if (i < C::WLDS / 4) ldsW[i] = wreg[k]; }
    }
  }
  ICS_TICK(8);
  // lane constants of the B operand: in window h this lane's 8 consecutive halves start at half
  // bo = 32h + 8*lg - li + 15 of the zero-padded row; it reads the five dwords that contain them and funnel-shifts
  // by the parity (v_alignbit).  (Gathering them from a row image with ds_bpermute cost ~5 LDS cycles per bpermute.)
  // wa0 = LDS byte address of the first dword in row 0; opaque to the optimiser so that the per-row constants stay in
  // the 16-bit offset field of the ds_read (folded with the plane base they exceed it: one address VGPR per row)
  typedef const __attribute__((address_space(3))) uint32_t* lds_u32p;
  uint32_t wa0[C::NCH], bsh[C::NCH];
#pragma unroll
  for (int h = 0; h < C::NCH; ++h) {
    const int bo = 32 * h + 8 * lg - li + 15;
    const bool bzero = bo < 8 || bo > K + 14;                       // window entirely in the zero padding
    bsh[h] = bzero ? 0u : (uint32_t)(bo & 1) * 16u;
    wa0[h] = (uint32_t)(uintptr_t)(lds_u32p)(lds + C::SCRATCH + 256) + 8u * (uint32_t)(bzero ? C::WZERO : ((bo - 8) >> 1));
    asm volatile("" : "+v"(wa0[h]));
  }
  const unsigned char* base_h = lds + li * C::ROWB + (16 * cb + 8 * lg) * 2;

  // Step-size maxima (pyx:523-524).  |g| and |u| are tracked as BIT PATTERNS under an unsigned integer maximum: for non-negative
  // floats that order is the float order, and every NaN pattern (> 0x7F800000) lies above +Inf, so a NaN propagates by itself like
  // np.amax does -- two instructions per value instead of the float maximum plus a compare / select / or for a separate flag.
  // max u (signed) stays a float maximum; its NaN shows in the |u| pattern.
  uint32_t mgb[3] = {0u, 0u, 0u}, mub[3] = {0u, 0u, 0u};
  float mu[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
  uint32_t rflags = 0u;   // bit 6: any element reduced

  // epilogue operands and output through buffer addressing as well (frame origins; offsets are >= 0 there)
  const __amdgpu_buffer_rsrc_t rs_o = make_rsrc(a.out);
  // (the first tile's rows were requested at the top of the kernel, ahead of the weight rows)

  int next_tile = 0;
#pragma unroll 1
  for (; tile < band1; tile = next_tile) {
    const int tyi = tile / tpr, txi = tile - tyi * tpr;
    const int x0 = TORG + txi * C::TW, y0 = TORG + tyi * C::TH;

    // ---- raw fp32 HWC rows (registers) -> six fp16 planes, scaled by a per-tile power of two ----------
    float inv_x;
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
      if (a.sched && tid == 0) lds_next[parity] = band0 + nx + (int)atomicAdd(a.sched + xcd, 1u);   // visible behind the second barrier
      __syncthreads();
#pragma unroll
      for (int w = 0; w < C::NW; ++w) m = __builtin_fmaxf(m, fscr[w]);
      float s_x;
      pow2_scale(__builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, m))), s_x, inv_x);   // uniform: SGPRs
      const int tidc = opaque(tid);
#pragma unroll
      for (int k = 0; k < C::NIT; ++k) {
        const int t = tidc + k * C::NT;
        if (t < C::NTASK && !((ICS_MFMA_ABLATE & 2) && k > 0)) {
          const int row = t / C::XG, xg = t - row * C::XG;
          const int rc = row % C::RS;
          const int crow = (rc == 0 ? 0 : (rc == 1 ? C::cls_base(1) : (rc == 2 ? C::cls_base(2) : C::cls_base(3)))) + row / C::RS;
          unsigned char* dst = lds + crow * C::ROWB + 8 * xg;
          float f[12];
#pragma unroll
          for (int h = 0; h < 3; ++h)
#pragma unroll
            for (int e = 0; e < 4; ++e) f[4 * h + e] = raw[k][h][e] * s_x;
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            h4 hi, lo;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
              const float x = f[3 * p + c];
              const _Float16 xh = (_Float16)x;
              hi[p] = xh;
              lo[p] = (_Float16)(x - (float)xh);   // (as one v_fma_mix from the raw value: 36 vector instructions fewer per tile, same time)
            }
            *reinterpret_cast<h4*>(dst + (2 * c) * C::PLANE) = hi;
            *reinterpret_cast<h4*>(dst + (2 * c + 1) * C::PLANE) = lo;
          }
        }
      }
    }
    __syncthreads();
    ICS_TICK(0);

    // ---- request the next tile's rows: in flight during the whole matrix phase (the loop below issues no
    // vector-memory loads -- they return in order, a weight load behind this prefetch would wait for it) -----
    // mode 0, 32-row tiles: the image operand of this tile's residual (accumulator order, ics_image_acc.h) is requested here, ahead
    // of the next tile's rows (loads return in order), and is there when the matrix phase ends
    constexpr bool EARLY = ICS_EPI_EARLY && MODE == 0 && RS == 2 && NH == 1;
    u4 fpre[3][C::RS];
    const float* faccp0 = (MODE == 0 && C::RS != 1) ? a.facc[C::RS == 2 ? 0 : 1] : nullptr;
    if (EARLY && faccp0 != nullptr && x0 + 16 * cb < xend) {
      const __amdgpu_buffer_rsrc_t rs_a0 = make_rsrc(faccp0);
      const int lv = 16 * (opaque(tid) & 63), sb0 = (tile * 4 + cb) * (3 * C::RS * 1024);
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int t = 0; t < C::RS; ++t) fpre[c][t] = __builtin_amdgcn_raw_buffer_load_b128(rs_a0, lv, sb0 + (c * C::RS + t) * 1024, ICS_EPI_LOAD_AUX0);
    }
    if (EARLY1 && !first1) request1(tile);   // (a workgroup's further tiles, if any: ahead of the matrix phase)
    first1 = false;
    next_tile = a.sched ? __builtin_amdgcn_readfirstlane(lds_next[parity]) : tile + nx;
    parity ^= 1;
    if (next_tile < band1) {
      const int nt = next_tile;
      const int nyi = nt / tpr, nxi = nt - nyi * tpr;
      load_raw<C>(raw, rs_in, 4 * ((a.g.ay + TORG + nyi * C::TH - C::PAD) * pitch + 3 * (a.g.ax + TORG + nxi * C::TW - C::PAD)), opaque(tid), pitch);
    }
    __builtin_amdgcn_sched_barrier(0);

    ICS_TICK(7);
    // ---- Toeplitz MFMA loop ----------------------------------------------------------------------------
    // (The two-window kernels, K >= 19, need the whole q loop unrolled for this -- the Makefile raises the pragma-unroll
    //  threshold for this file; with the default threshold the loop stayed rolled, the B fragments went to scratch and
    //  the kernel was 5x slower: 3.3 vs 0.51 ms at 4096^2 / 31x31.)
    constexpr bool INTERLEAVE = ICS_MFMA_INTERLEAVE != 0;
    f4 acc[3][C::RS];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch)
#pragma unroll
      for (int t = 0; t < C::RS; ++t) acc[ch][t] = (f4){0.f, 0.f, 0.f, 0.f};
    // kernel rows [A0, A1) of the three channels: the whole PSF (NH = 1) or this wave's half of it
    auto matrix_phase = [&](auto a0c, auto a1c) {
      constexpr int A0 = decltype(a0c)::value, A1 = decltype(a1c)::value;
      constexpr int Q0 = A0, Q1 = A1 + C::RS - 1;   // fragments q = Q0 .. Q1 - 1 meet these rows
#pragma unroll
      for (int ch = 0; ch < ((ICS_MFMA_ABLATE & 1) ? 0 : 3); ++ch) {
        // (re-hidden per tile and channel: the weight reads are tile-invariant and would otherwise be hoisted out
        //  of the tile loop, hundreds of live registers)
        uint32_t wb[C::NCH];
#pragma unroll
        for (int h = 0; h < C::NCH; ++h) { wb[h] = wa0[h]; asm volatile("" : "+v"(wb[h])); }
        const unsigned char* ph = base_h + (2 * ch) * C::PLANE;
        const unsigned char* pl = ph + C::PLANE;
        h8 Bh[K][C::NCH], Bl[K][C::NCH];
        // software pipeline: the operands of step q + 1 (A fragments from the planes, B fragments from the weight
        // rows) are requested before the MFMAs of step q
        // B fragments in two halves: the raw dwords of kernel row q + 1 are requested before the MFMAs of step q and
        // funnel-shifted behind them.  The reads are volatile: as plain loads they were sunk to the shifts, and the wave
        // waited out the LDS latency in front of every step's MFMAs (466 cycles per step for 192 cycles of MFMA,
        // tools/bench_conv_mfma.hip -DICS_MFMA_TRACE).
        typedef uint32_t u2 __attribute__((ext_vector_type(2)));
        typedef const volatile __attribute__((address_space(3))) u2* lds_vu2p;
        u2 rawB[C::NCH][5];   // (hi, lo) dword pairs: the table interleaves the two split terms dword by dword
        auto issueB = [&](int ka) {
#pragma unroll
          for (int h = 0; h < C::NCH; ++h) {
            const lds_vu2p r = reinterpret_cast<lds_vu2p>(wb[h] + (uint32_t)((ch * K + ka) * 2 * C::WROWB));
#pragma unroll
            for (int d = 0; d < 5; ++d) rawB[h][d] = (ICS_MFMA_ABLATE & 32) ? (u2){0x3c003c00u + ka + d, 0x3c003c00u + d} : r[d];
          }
        };
        auto finishB = [&](int ka) {
#pragma unroll
          for (int h = 0; h < C::NCH; ++h) {
            const u2* d = rawB[h];
            u4 wh = {__builtin_amdgcn_alignbit(d[1].x, d[0].x, bsh[h]), __builtin_amdgcn_alignbit(d[2].x, d[1].x, bsh[h]),
                     __builtin_amdgcn_alignbit(d[3].x, d[2].x, bsh[h]), __builtin_amdgcn_alignbit(d[4].x, d[3].x, bsh[h])};
            u4 wl = {__builtin_amdgcn_alignbit(d[1].y, d[0].y, bsh[h]), __builtin_amdgcn_alignbit(d[2].y, d[1].y, bsh[h]),
                     __builtin_amdgcn_alignbit(d[3].y, d[2].y, bsh[h]), __builtin_amdgcn_alignbit(d[4].y, d[3].y, bsh[h])};
            Bh[ka][h] = __builtin_bit_cast(h8, wh);
            Bl[ka][h] = __builtin_bit_cast(h8, wl);
          }
        };
        issueB(A0);
        h8 Ah[C::NCH], Al[C::NCH];
        {
          constexpr int off0 = (C::cls_base(Q0 % C::RS) + Q0 / C::RS) * C::ROWB;
#pragma unroll
          for (int h = 0; h < C::NCH; ++h) { Ah[h] = *reinterpret_cast<const h8*>(ph + off0 + 64 * h); Al[h] = *reinterpret_cast<const h8*>(pl + off0 + 64 * h); }
        }
        finishB(A0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = Q0; q < Q1; ++q) {
          h8 Nh[C::NCH], Nl[C::NCH];
#pragma unroll
          for (int h = 0; h < C::NCH; ++h) { Nh[h] = Ah[h]; Nl[h] = Al[h]; }
          if (q + 1 < Q1) {
            const int off = (C::cls_base((q + 1) % C::RS) + (q + 1) / C::RS) * C::ROWB;
            if (!(ICS_MFMA_ABLATE & 16)) {   // 16: timing probe without the A-fragment reads
#pragma unroll
              for (int h = 0; h < C::NCH; ++h) {
                Nh[h] = *reinterpret_cast<const h8*>(ph + off + 64 * h);
                Nl[h] = *reinterpret_cast<const h8*>(pl + off + 64 * h);
              }
            }
          }
          if (q + 1 < A1) issueB(q + 1);
          if (!INTERLEAVE) __builtin_amdgcn_sched_barrier(0);   // ...all requested before the step's MFMAs start
          // three split terms x windows; the (up to) 4 accumulators of a pass are independent
#pragma unroll
          for (int term = 0; term < 3; ++term) {
#pragma unroll
            for (int h = 0; h < C::NCH; ++h) {
#pragma unroll
              for (int t = 0; t < C::RS; ++t) {
                const int ka = q - t;
                if (ka < A0 || ka >= A1) continue;
                if ((ICS_MFMA_ABLATE & 64) && term == 2) continue;   // timing probe: a third of the MFMAs less (results wrong)
                const h8 av = term == 2 ? Al[h] : Ah[h];
                const h8 bv = term == 1 ? Bl[ka][h] : Bh[ka][h];
                acc[ch][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bv, acc[ch][t], 0, 0, 0);
              }
            }
          }
          if (!INTERLEAVE) __builtin_amdgcn_sched_barrier(0);   // ...and consumed behind them
          if (q + 1 < A1) finishB(q + 1);
#pragma unroll
          for (int h = 0; h < C::NCH; ++h) { Ah[h] = Nh[h]; Al[h] = Nl[h]; }
          // issue order inside the step: the LDS reads go into the shadows of the first MFMAs (an MFMA holds the matrix
          // pipe for 16 cycles, the wave can issue an independent instruction meanwhile), the funnel shifts into the
          // shadows of the last ones; as three blocks (reads | MFMAs | shifts) a step cost their sum
          if (INTERLEAVE) {
            int nt = 0;
#pragma unroll
            for (int t = 0; t < C::RS; ++t) nt += (q - t >= A0 && q - t < A1) ? 1 : 0;
            const int nm = 3 * C::NCH * nt;                                             // MFMAs of this step
            const int nr = ((q + 1 < Q1) ? 2 * C::NCH : 0) + ((q + 1 < A1) ? 5 * C::NCH : 0);   // LDS reads
            const int nv = (q + 1 < A1) ? 8 * C::NCH : 0;                               // funnel shifts
            const int tail = nv ? (nm > 4 ? 4 : nm) : 0;                                // MFMAs that cover the shifts
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
          __builtin_amdgcn_sched_barrier(0);   // keep each step's prefetches in that step (register pressure)
        }
      }
    };
    // MCfg::PAIR: accumulator sets H and H + 2 of this wave, all kernel rows: pairs (2s, 2s + 1), s < NP, then the single row K - 1
    // (classic two windows).  Step s' works on the fragments f = 2s' + H, f + 1: set H takes item s', set H + 2 item s' - 1 (same
    // fragments, since (2s' - 2) + (H + 2) = f), so the B fragments of an item are built once and used in two consecutive steps.
    auto matrix_phase_pairs = [&](auto hc) {
      constexpr int H = decltype(hc)::value;
      constexpr int NP = (K - 1) / 2;
      typedef uint32_t u2 __attribute__((ext_vector_type(2)));
      typedef const volatile __attribute__((address_space(3))) u2* lds_vu2p;
      // B windows of a lane: start half bo of the zero-padded weight row (as in the classic loop), four kinds:
      //   0: Toeplitz rows [0, 32) of row a      1: rows [32, 64) of row a (single row only)      2: rows [16, 48) of row a + 1
      //   3 (mixed): lane groups 0, 1 rows [32, 48) of row a, lane groups 2, 3 rows [0, 16) of row a + 1
      const uint32_t wbase = (uint32_t)(uintptr_t)(lds_u32p)(lds + C::SCRATCH + 256);
      auto wof = [&](int bo, int rowadd) -> uint32_t {
        const bool z = bo < 8 || bo > K + 14;
        return wbase + 8u * (uint32_t)(z ? C::WZERO : ((bo - 8) >> 1)) + (z ? 0u : (uint32_t)(rowadd * 2 * C::WROWB));
      };
      uint32_t wk[4] = {wof(8 * lg - li + 15, 0), wof(32 + 8 * lg - li + 15, 0), wof(16 + 8 * lg - li + 15, 0),
                        lg < 2 ? wof(32 + 8 * lg - li + 15, 0) : wof(8 * (lg - 2) - li + 15, 1)};
      const uint32_t sh = (uint32_t)((8 * lg - li + 15) & 1) * 16u;   // (every start above has this parity)
      // A fragments: `base_h` = lane row li, columns 16 cb + 8 lg of LDS row 0.  The mixed fragment's lane groups 2, 3 sit 16
      // columns to the left in the NEXT input row: the distance between the LDS rows of fragments f and f + 1 depends on f mod 4
      // only (row classes), and f mod 4 is H or H + 2 here -- two per-lane offsets
      constexpr int DA = (C::cls_base((H + 1) % 4) + (H + 1) / 4 - C::cls_base(H % 4) - H / 4) * C::ROWB;
      constexpr int DB = (C::cls_base((H + 3) % 4) + (H + 3) / 4 - C::cls_base((H + 2) % 4) - (H + 2) / 4) * C::ROWB;
      int mixA = lg < 2 ? 64 : DA - 32, mixB = lg < 2 ? 64 : DB - 32;
      asm volatile("" : "+v"(mixA), "+v"(mixB));
#pragma unroll
      for (int ch = 0; ch < ((ICS_MFMA_ABLATE & 1) ? 0 : 3); ++ch) {
        uint32_t wb[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { wb[k] = wk[k]; asm volatile("" : "+v"(wb[k])); }
        const unsigned char* ph = base_h + (2 * ch) * C::PLANE;
        const unsigned char* pl = ph + C::PLANE;
        // item i < NP: pair (2i, 2i + 1) -> kinds 0, 3, 2; item NP: single row K - 1 -> kinds 0, 1
        u2 rawB[3][5];
        h8 Bc[3][2], Bp[3][2];                       // [window][hi, lo] of the current item (set H) and of the previous one (set H + 2)
        auto issueB = [&](int item) {
          const int a = 2 * item;
          const int kinds[3] = {0, item < NP ? 3 : 1, 2};
#pragma unroll
          for (int w = 0; w < (item < NP ? 3 : 2); ++w) {
            const int arow = a + (w == 2 ? 1 : 0);   // (the mixed window adds its second row per lane, in wk[3])
            const lds_vu2p r = reinterpret_cast<lds_vu2p>(wb[kinds[w]] + (uint32_t)((ch * K + arow) * 2 * C::WROWB));
#pragma unroll
            for (int d = 0; d < 5; ++d) rawB[w][d] = r[d];
          }
        };
        auto finishB = [&](int item, h8 (&B)[3][2]) {
#pragma unroll
          for (int w = 0; w < (item < NP ? 3 : 2); ++w) {
            const u2* d = rawB[w];
            u4 wh = {__builtin_amdgcn_alignbit(d[1].x, d[0].x, sh), __builtin_amdgcn_alignbit(d[2].x, d[1].x, sh),
                     __builtin_amdgcn_alignbit(d[3].x, d[2].x, sh), __builtin_amdgcn_alignbit(d[4].x, d[3].x, sh)};
            u4 wl = {__builtin_amdgcn_alignbit(d[1].y, d[0].y, sh), __builtin_amdgcn_alignbit(d[2].y, d[1].y, sh),
                     __builtin_amdgcn_alignbit(d[3].y, d[2].y, sh), __builtin_amdgcn_alignbit(d[4].y, d[3].y, sh)};
            B[w][0] = __builtin_bit_cast(h8, wh);
            B[w][1] = __builtin_bit_cast(h8, wl);
          }
        };
        // A fragments of step s': [0] row f cols [0, 32), [1] mixed (item < NP) or row f cols [32, 64) (single), [2] row f + 1 cols [16, 48)
        auto loadA = [&](int sp, h8 (&A)[3][2], bool need_pair, bool need_single, h8 (&A1)[2]) {
          const int f = 2 * sp + H;
          const int off = (C::cls_base(f % 4) + f / 4) * C::ROWB;
          const int off1 = (C::cls_base((f + 1) % 4) + (f + 1) / 4) * C::ROWB;
          A[0][0] = *reinterpret_cast<const h8*>(ph + off); A[0][1] = *reinterpret_cast<const h8*>(pl + off);
          if (need_pair) {
            const int mix = (f % 4 == H) ? mixA : mixB;
            A[1][0] = *reinterpret_cast<const h8*>(ph + off + mix); A[1][1] = *reinterpret_cast<const h8*>(pl + off + mix);
            A[2][0] = *reinterpret_cast<const h8*>(ph + off1 + 32); A[2][1] = *reinterpret_cast<const h8*>(pl + off1 + 32);
          }
          if (need_single) { A1[0] = *reinterpret_cast<const h8*>(ph + off + 64); A1[1] = *reinterpret_cast<const h8*>(pl + off + 64); }
        };
        // which items run at step sp: set H item sp (pair if sp < NP, single if sp == NP), set H + 2 item sp - 1
        auto pair_at = [](int sp) { return sp < NP || (sp >= 1 && sp - 1 < NP); };
        auto single_at = [](int sp) { return sp == NP || sp - 1 == NP; };
        h8 Ac[3][2], As[2];
        issueB(0);
        loadA(0, Ac, pair_at(0), single_at(0), As);
        finishB(0, Bc);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int sp = 0; sp <= NP + 1; ++sp) {
          h8 An[3][2], Asn[2];
#pragma unroll
          for (int w = 0; w < 3; ++w) { An[w][0] = Ac[w][0]; An[w][1] = Ac[w][1]; }
          Asn[0] = As[0]; Asn[1] = As[1];
          if (sp + 1 <= NP + 1) loadA(sp + 1, An, pair_at(sp + 1), single_at(sp + 1), Asn);
          if (sp + 1 <= NP) issueB(sp + 1);
          int nm = 0;
          // three split terms; the two sets' accumulators alternate
#pragma unroll
          for (int term = 0; term < 3; ++term) {
            const int ia = term == 2 ? 1 : 0, ib = term == 1 ? 1 : 0;
#pragma unroll
            for (int w = 0; w < 3; ++w) {
              if ((ICS_MFMA_ABLATE & 64) && term == 2) continue;
              // set H: item sp
              if (sp < NP) { acc[ch][H] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ac[w][ia], Bc[w][ib], acc[ch][H], 0, 0, 0); ++nm; }
              else if (sp == NP && w < 2) { acc[ch][H] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w == 0 ? Ac[0][ia] : As[ia], Bc[w][ib], acc[ch][H], 0, 0, 0); ++nm; }
              // set H + 2: item sp - 1
              if (sp >= 1 && sp - 1 < NP) { acc[ch][H + 2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ac[w][ia], Bp[w][ib], acc[ch][H + 2], 0, 0, 0); ++nm; }
              else if (sp - 1 == NP && w < 2) { acc[ch][H + 2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w == 0 ? Ac[0][ia] : As[ia], Bp[w][ib], acc[ch][H + 2], 0, 0, 0); ++nm; }
            }
          }
#pragma unroll
          for (int w = 0; w < 3; ++w) { Bp[w][0] = Bc[w][0]; Bp[w][1] = Bc[w][1]; }
          if (sp + 1 <= NP) finishB(sp + 1, Bc);
#pragma unroll
          for (int w = 0; w < 3; ++w) { Ac[w][0] = An[w][0]; Ac[w][1] = An[w][1]; }
          As[0] = Asn[0]; As[1] = Asn[1];
          if (ICS_MFMA_INTERLEAVE) {   // LDS reads behind the first MFMAs of the step, the funnel shifts behind the last four
            int nr = 0, nv = 0;
            if (sp + 1 <= NP + 1) nr += 2 + (pair_at(sp + 1) ? 4 : 0) + (single_at(sp + 1) ? 2 : 0);
            if (sp + 1 <= NP) { nr += 5 * (sp + 1 < NP ? 3 : 2); nv = 8 * (sp + 1 < NP ? 3 : 2); }
            const int tail = nv ? (nm > 4 ? 4 : nm) : 0, head = nm - tail;
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
      }
    };
    // MCfg::PAIR_ITEMS: this wave takes the items [I0, I1] (item i < NP = row pair (2i, 2i + 1), item NP = the single row K - 1) for
    // ALL four accumulator sets.  Fragment step f serves the sets t = f mod 2 and t + 2 with the items (f - t) / 2; an item's B
    // fragments are built once and used in the four steps f = 2i .. 2i + 3 -- half the weight-row reads and funnel shifts per
    // MFMA of the per-set split above (which ran at the wave's issue limit: 2.8 other instructions per MFMA, 1.00 of 1.06 ms).
    auto matrix_phase_items = [&](auto i0c, auto i1c) {
      constexpr int I0 = decltype(i0c)::value, I1 = decltype(i1c)::value;
      constexpr int NP = (K - 1) / 2;
      typedef uint32_t u2 __attribute__((ext_vector_type(2)));
      typedef const volatile __attribute__((address_space(3))) u2* lds_vu2p;
      const uint32_t wbase = (uint32_t)(uintptr_t)(lds_u32p)(lds + C::SCRATCH + 256);
      auto wof = [&](int bo, int rowadd) -> uint32_t {
        const bool z = bo < 8 || bo > K + 14;
        if constexpr (C::WSPLIT)   // byte address of the first of the lane's four hi dwords: copy of its parity, dword (bo - 8 - parity) / 2
          return wbase + 4u * (uint32_t)(z ? C::WZERO : ((bo & 1) * 2 * (C::WROWB / 4) + ((bo - 8 - (bo & 1)) >> 1))) + (z ? 0u : (uint32_t)(rowadd * 4 * C::WROWB));
        else
          return wbase + 8u * (uint32_t)(z ? C::WZERO : ((bo - 8) >> 1)) + (z ? 0u : (uint32_t)(rowadd * 2 * C::WROWB));
      };
      uint32_t wk[4] = {wof(8 * lg - li + 15, 0), wof(32 + 8 * lg - li + 15, 0), wof(16 + 8 * lg - li + 15, 0),
                        lg < 2 ? wof(32 + 8 * lg - li + 15, 0) : wof(8 * (lg - 2) - li + 15, 1)};
      const uint32_t sh = (uint32_t)((8 * lg - li + 15) & 1) * 16u;
      // mixed A fragment: per-lane offset from fragment f's row to (lane groups 2, 3) fragment f + 1's row, 16 columns to the left; the
      // row distance depends on f mod 4
      // (one lane mask and a constant per f mod 4 instead of four per-lane offsets: three registers less in a 256-register kernel)
      int mixsel = lg < 2 ? 0 : -1;
      asm volatile("" : "+v"(mixsel));
      auto mixoff = [&](int c4) { return 64 + (mixsel & ((C::cls_base((c4 + 1) % 4) + (c4 + 1) / 4 - C::cls_base(c4)) * C::ROWB - 96)); };
      constexpr int F0 = 2 * I0, F1 = 2 * I1 + 3;            // fragment steps of this wave
#pragma unroll
      for (int ch = 0; ch < ((ICS_MFMA_ABLATE & 1) ? 0 : 3); ++ch) {
        uint32_t wb[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { wb[k] = wk[k]; asm volatile("" : "+v"(wb[k])); }
        const unsigned char* ph = base_h + (2 * ch) * C::PLANE;
        const unsigned char* pl = ph + C::PLANE;
        u2 rawB[3][5];
        h8 Bn[3][2];                                           // (WSPLIT) the fragments just requested
        h8 Bs[2][3][2];                                        // [item & 1][window][hi, lo]
        auto issueB = [&](int item) {
          const int a = 2 * item;
          const int kinds[3] = {0, item < NP ? 3 : 1, 2};
#pragma unroll
          for (int w = 0; w < (item < NP ? 3 : 2); ++w) {
            const int arow = a + (w == 2 ? 1 : 0);
            if constexpr (C::WSPLIT) {
              // (as C++ loads the four 8-byte reads were merged into ds_read2_b64 at 4-byte alignment -- legal in the unaligned access mode
              //  of gfx950 and 2.8x slower for the whole kernel; ds_read2_b32 is what the alignment allows.  Inline asm results are not
              //  tracked by the compiler's s_waitcnt insertion: finishB() waits for them)
              const uint32_t ad = wb[kinds[w]] + (uint32_t)((ch * K + arow) * 4 * C::WROWB);
              constexpr int RD = C::WROWB / 4;
              u2 h01, h23, l01, l23;
              asm volatile("ds_read2_b32 %0, %1 offset1:1" : "=v"(h01) : "v"(ad));
              asm volatile("ds_read2_b32 %0, %1 offset0:2 offset1:3" : "=v"(h23) : "v"(ad));
              asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(l01) : "v"(ad), "n"(RD), "n"(RD + 1));
              asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(l23) : "v"(ad), "n"(RD + 2), "n"(RD + 3));
              Bn[w][0] = __builtin_bit_cast(h8, (u4){h01.x, h01.y, h23.x, h23.y});   // (into fresh registers: the slot of this item is
              Bn[w][1] = __builtin_bit_cast(h8, (u4){l01.x, l01.y, l23.x, l23.y});   //  still read by the MFMAs of the step that requests it)
            } else {
              const lds_vu2p r = reinterpret_cast<lds_vu2p>(wb[kinds[w]] + (uint32_t)((ch * K + arow) * 2 * C::WROWB));
#pragma unroll
              for (int d = 0; d < 5; ++d) rawB[w][d] = r[d];
            }
          }
        };
        auto finishB = [&](int item) {
          if constexpr (C::WSPLIT) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int w = 0; w < 3; ++w) { Bs[item & 1][w][0] = Bn[w][0]; Bs[item & 1][w][1] = Bn[w][1]; }
            return;
          }
#pragma unroll
          for (int w = 0; w < (item < NP ? 3 : 2); ++w) {
            const u2* d = rawB[w];
            u4 wh = {__builtin_amdgcn_alignbit(d[1].x, d[0].x, sh), __builtin_amdgcn_alignbit(d[2].x, d[1].x, sh),
                     __builtin_amdgcn_alignbit(d[3].x, d[2].x, sh), __builtin_amdgcn_alignbit(d[4].x, d[3].x, sh)};
            u4 wl = {__builtin_amdgcn_alignbit(d[1].y, d[0].y, sh), __builtin_amdgcn_alignbit(d[2].y, d[1].y, sh),
                     __builtin_amdgcn_alignbit(d[3].y, d[2].y, sh), __builtin_amdgcn_alignbit(d[4].y, d[3].y, sh)};
            Bs[item & 1][w][0] = __builtin_bit_cast(h8, wh);
            Bs[item & 1][w][1] = __builtin_bit_cast(h8, wl);
          }
        };
        // what step f needs: a pair item for some set -> the fragment triple; the single -> windows 0 and 1 of fragment f
        auto item_of = [](int f, int t) { return (f - t) / 2; };
        auto active = [&](int f, int t) { return f - t >= 0 && item_of(f, t) >= I0 && item_of(f, t) <= I1; };
        auto pair_at = [&](int f) { const int t = f & 1; return (active(f, t) && item_of(f, t) < NP) || (active(f, t + 2) && item_of(f, t + 2) < NP); };
        auto single_at = [&](int f) { const int t = f & 1; return (active(f, t) && item_of(f, t) == NP) || (active(f, t + 2) && item_of(f, t + 2) == NP); };
        auto loadA = [&](int f, h8 (&A)[3][2], h8 (&A1)[2]) {
          const int off = (C::cls_base(f % 4) + f / 4) * C::ROWB;
          const int off1 = (C::cls_base((f + 1) % 4) + (f + 1) / 4) * C::ROWB;
          A[0][0] = *reinterpret_cast<const h8*>(ph + off); A[0][1] = *reinterpret_cast<const h8*>(pl + off);
          if (pair_at(f)) {
            const int m = mixoff(f % 4);
            A[1][0] = *reinterpret_cast<const h8*>(ph + off + m); A[1][1] = *reinterpret_cast<const h8*>(pl + off + m);
            A[2][0] = *reinterpret_cast<const h8*>(ph + off1 + 32); A[2][1] = *reinterpret_cast<const h8*>(pl + off1 + 32);
          }
          if (single_at(f)) { A1[0] = *reinterpret_cast<const h8*>(ph + off + 64); A1[1] = *reinterpret_cast<const h8*>(pl + off + 64); }
        };
        h8 Ac[3][2], As[2];
        issueB(I0);
        loadA(F0, Ac, As);
        finishB(I0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int f = F0; f <= F1; ++f) {
          h8 An[3][2], Asn[2];
#pragma unroll
          for (int w = 0; w < 3; ++w) { An[w][0] = Ac[w][0]; An[w][1] = Ac[w][1]; }
          Asn[0] = As[0]; Asn[1] = As[1];
          if (f + 1 <= F1) loadA(f + 1, An, Asn);
          // item (f + 1) / 2 starts at the even step f + 1: its weight rows are requested here, shifted behind this step's MFMAs
          const bool newB = ((f + 1) & 1) == 0 && (f + 1) / 2 > I0 && (f + 1) / 2 <= I1;
          if (newB) issueB((f + 1) / 2);
          int nm = 0;
#pragma unroll
          for (int term = 0; term < 3; ++term) {
            const int ia = term == 2 ? 1 : 0, ib = term == 1 ? 1 : 0;
#pragma unroll
            for (int w = 0; w < 3; ++w) {
              if ((ICS_MFMA_ABLATE & 64) && term == 2) continue;
#pragma unroll
              for (int tt = 0; tt < 2; ++tt) {
                const int t = (f & 1) + 2 * tt;
                if (!active(f, t)) continue;
                const int it = item_of(f, t);
                if (it < NP) { acc[ch][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ac[w][ia], Bs[it & 1][w][ib], acc[ch][t], 0, 0, 0); ++nm; }
                else if (w < 2) { acc[ch][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w == 0 ? Ac[0][ia] : As[ia], Bs[it & 1][w][ib], acc[ch][t], 0, 0, 0); ++nm; }
              }
            }
          }
          if (newB) finishB((f + 1) / 2);
#pragma unroll
          for (int w = 0; w < 3; ++w) { Ac[w][0] = An[w][0]; Ac[w][1] = An[w][1]; }
          As[0] = Asn[0]; As[1] = Asn[1];
          if (ICS_MFMA_INTERLEAVE) {
            int nr = 0, nv = 0;
            if (f + 1 <= F1) nr += 2 + (pair_at(f + 1) ? 4 : 0) + (single_at(f + 1) ? 2 : 0);
            if (newB) { const int nw = (f + 1) / 2 < NP ? 3 : 2; nr += (C::WSPLIT ? 4 : 5) * nw; nv = C::WSPLIT ? 0 : 8 * nw; }
            const int tail = nv ? (nm > 4 ? 4 : nm) : 0, head = nm - tail;
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
      }
    };
    const bool has_out = x0 + 16 * cb < xend;   // wave-uniform: a column block right of the output carries none
    if (has_out) {
      if constexpr (C::PAIR_ITEMS) {
        constexpr int NPI = (K - 1) / 2, ISPLIT = (NPI + 1) / 2;          // items 0 .. ISPLIT - 1 | ISPLIT .. NPI (the single row included)
        if (half == 0) matrix_phase_items(std::integral_constant<int, 0>{}, std::integral_constant<int, ISPLIT - 1>{});
        else matrix_phase_items(std::integral_constant<int, ISPLIT>{}, std::integral_constant<int, NPI>{});
      } else if constexpr (C::PAIR_SETS) {
        if (half == 0) matrix_phase_pairs(std::integral_constant<int, 0>{}); else matrix_phase_pairs(std::integral_constant<int, 1>{});
      } else {
        if (NH == 1) matrix_phase(std::integral_constant<int, 0>{}, std::integral_constant<int, K>{});
        else if (half == 0) matrix_phase(std::integral_constant<int, 0>{}, std::integral_constant<int, C::KSPLIT>{});
        else matrix_phase(std::integral_constant<int, C::KSPLIT>{}, std::integral_constant<int, K>{});
      }
    }
    if (NH == 2 && !C::PAIR_SETS) {
      // partial sums of the two halves: a wave keeps the accumulator sets t = 2 half, 2 half + 1 and hands the other two to its
      // partner through the plane space (free once every wave has left the matrix phase)
      lds_barrier();
      f4* xch = reinterpret_cast<f4*>(lds);
      auto exchange = [&](auto hc, bool write) {
        constexpr int H = decltype(hc)::value;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch)
#pragma unroll
          for (int tt = 0; tt < C::RS / 2; ++tt) {
            if (write) xch[(((cb * 2 + (1 - H)) * 3 + ch) * (C::RS / 2) + tt) * 64 + lane] = acc[ch][(C::RS / 2) * (1 - H) + tt];
            else acc[ch][(C::RS / 2) * H + tt] += xch[(((cb * 2 + H) * 3 + ch) * (C::RS / 2) + tt) * 64 + lane];
          }
      };
      if (half == 0) exchange(std::integral_constant<int, 0>{}, true); else exchange(std::integral_constant<int, 1>{}, true);
      lds_barrier();
      if (half == 0) exchange(std::integral_constant<int, 0>{}, false); else exchange(std::integral_constant<int, 1>{}, false);
    }
    ICS_TICK(1);
    // ---- epilogue straight from the accumulators: lane (li, lg) holds, for each channel, the 16 rows
    // t + 16*lg + 4*r of pixel column 16*wv + li, i.e. one 12-byte HWC pixel per (t, r): operands arrive and
    // results leave as dwordx3 (16 lanes = 192 contiguous bytes of a row), no LDS transpose and no workgroup
    // barrier between the matrix phase and the stores -- the four waves drift apart and overlap their phases.
    if (has_out && !(ICS_MFMA_ABLATE & 4)) {
      const float sc = inv_w * inv_x;   // powers of two
      const int tide = opaque(tid);
      const int eli = tide & 15, elg = (tide >> 4) & 3;
      const int colx = x0 + 16 * cb + eli;
      const int voff = 4 * (4 * C::RS * elg * pitch + 3 * eli);   // lane part of the byte offset
      const int sb = 4 * (y0 * pitch + 3 * (x0 + 16 * cb));      // wave-uniform part (tile origin + column block)
      constexpr int EOPS = (MODE == 0) ? 1 : 2;
      // TB = accumulator sets per batch: the operands of a batch are requested together.  (Measured without gain, and removed:
      // requesting the operands before the matrix phase -- the kernel follows its memory traffic, not the epilogue's latency --
      // and a single-operand variant for the launches where the majoriser frame is the u frame.)
      // TVOP (extended modes): the T frame is a third operand, requested like the others (as per-element scalar loads it
      // cost the back-projection +40 %)
      const __amdgpu_buffer_rsrc_t rs_tv = make_rsrc(a.tv);
      const float* faccp = (MODE == 0 && C::RS != 1) ? a.facc[C::RS == 2 ? 0 : 1] : nullptr;
      const bool use_acc = MODE == 0 && NH == 1 && faccp != nullptr;           // uniform
      const __amdgpu_buffer_rsrc_t rs_acc = make_rsrc(faccp);
      const int acc_voff = 16 * (tide & 63);
      const int acc_sb = (tile * 4 + cb) * (3 * C::RS * 1024);                  // bytes: [tile][cb][ch][t][lane] float4
      auto run_epi = [&](auto tbc, auto tvc, auto tloc) {
      // accumulator sets of this wave: all of them (NH = 1), two consecutive ones (NH = 2), or TLO and TLO + 2 (row pairs, MCfg::PAIR).
      // The loops below run over slots i = 0 .. NS - 1 <-> set TLO + i * TSTEP; written with `t` running over [TLO, THI) in steps of TSTEP.
      constexpr int TSTEP = C::PAIR_SETS ? 2 : 1, NS = C::RS / NH;
      constexpr int TLO = decltype(tloc)::value, THI = TLO + NS * TSTEP;
      constexpr int TB = (decltype(tbc)::value < NS ? decltype(tbc)::value : NS) * TSTEP;   // sets per batch, in units of t
      constexpr bool TVOP = decltype(tvc)::value;
      // PAM kinds (tv_kind 2, 3): the frame this kernel writes is G = T + lambd * gradu itself -- the update pass then reads u and G
      // only (no T, no majoriser, no image: 3 frame transits instead of 5), and T is read exactly once, here
      const bool pam = TVOP && a.tv_kind >= 2;   // uniform
#pragma unroll
      for (int t0 = TLO; t0 < THI; t0 += TB) {
      u3 eop[EOPS][C::RS][4], eopT[C::RS][4];
#pragma unroll
      for (int t = t0; t < t0 + TB; t += TSTEP)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int so = sb + 4 * (t + C::RS * r) * pitch;
          if (ICS_MFMA_ABLATE & 8) { eop[0][t][r] = (u3){0u, 0u, 0u}; eop[EOPS - 1][t][r] = (u3){0u, 0u, 0u}; continue; }
          if (MODE == 0 && use_acc) continue;   // requested below, per (channel, t)
          if (EARLY1) { eop[0][t][r] = pre1[r]; continue; }
          eop[0][t][r] = __builtin_amdgcn_raw_buffer_load_b96(rs_f, voff, so, (MODE == 0 ? ICS_EPI_LOAD_AUX0 : ICS_EPI_LOAD_AUX));
          if (MODE == 1 && !pam) eop[EOPS - 1][t][r] = __builtin_amdgcn_raw_buffer_load_b96(rs_t, voff, so, ICS_EPI_LOAD_AUX);   // (PAM has no majoriser term)
          if (MODE == 1 && TVOP) eopT[t][r] = __builtin_amdgcn_raw_buffer_load_b96(rs_tv, voff, so, ICS_EPI_LOAD_AUX);
        }
      if (MODE == 0 && use_acc && !(ICS_MFMA_ABLATE & 8)) {
        // the image in accumulator order (ics_image_acc.h): one 16-byte load per (channel, accumulator set) instead of four 12-byte ones
#pragma unroll
        for (int t = t0; t < t0 + TB; t += TSTEP)
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const u4 v = EARLY ? fpre[c][t] : __builtin_amdgcn_raw_buffer_load_b128(rs_acc, acc_voff, acc_sb + (c * C::RS + t) * 1024, ICS_EPI_LOAD_AUX0);
#pragma unroll
            for (int r = 0; r < 4; ++r) eop[0][t][r][c] = v[r];
          }
      }
      if (t0 == TLO) ICS_TICK(3);
      if (MODE == 1 && t0 == TLO && !pam) {
        // the back-projection itself needs no operand: all 16 rows are stored behind the first batch of requests, in the
        // shadow of their latency (u and ut only feed the step-size reductions)
#pragma unroll
        for (int t = TLO; t < THI; t += TSTEP)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int y = y0 + t + 4 * C::RS * elg + C::RS * r;
            if (y < a.g.uM && colx < a.g.uN && !(ICS_MFMA_ABLATE & 8)) {
              const u3 e = {__float_as_uint(acc[0][t][r] * sc), __float_as_uint(acc[1][t][r] * sc), __float_as_uint(acc[2][t][r] * sc)};
              __builtin_amdgcn_raw_buffer_store_b96(e, rs_o, voff, sb + 4 * (t + C::RS * r) * pitch, ICS_EPI_STORE_AUX);
            }
          }
      }
#pragma unroll
      for (int t = t0; t < t0 + TB; t += TSTEP)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int y = y0 + t + 4 * C::RS * elg + C::RS * r;
          const int so = sb + 4 * (t + C::RS * r) * pitch;
          float av[3];
#pragma unroll
          for (int c = 0; c < 3; ++c) av[c] = acc[c][t][r] * sc;
          if (MODE == 0) {
            // error = synth - image on the M x N interior (pyx:488); the border ring of the frame stays 0
            if (y >= C::PAD && y < C::PAD + a.g.M && colx >= C::PAD && colx < C::PAD + a.g.N && (!(ICS_MFMA_ABLATE & 8) || av[0] + av[1] + av[2] == 12345.678f)) {
              u3 e;
#pragma unroll
              for (int c = 0; c < 3; ++c) e[c] = __float_as_uint(__fsub_rn(av[c], __uint_as_float(eop[0][t][r][c])));
              __builtin_amdgcn_raw_buffer_store_b96(e, rs_o, voff, so, ICS_EPI_STORE_AUX);
            }
          } else {
            // gradu over the whole u-frame + reductions for the step size (pyx:519,523-524)
            if (y < a.g.uM && colx < a.g.uN && (!(ICS_MFMA_ABLATE & 8) || av[0] + av[1] + av[2] == 12345.678f)) {
              const float lambd = a.lambd;
              u3 gout;
#pragma unroll
              for (int c = 0; c < 3; ++c) {
                const float uv = __uint_as_float(eop[0][t][r][c]), tv = pam ? 0.f : __uint_as_float(eop[EOPS - 1][t][r][c]);
                float g;
                const float Tv = TVOP ? __uint_as_float(eopT[t][r][c]) : 0.f;
                if (TVOP && a.tv_kind >= 2)
                  g = (float)((double)Tv + (double)__fmul_rn(lambd, av[c]));
                else if (TVOP && a.tv_kind == 1 && y >= 1 && y <= a.g.uM - 2 && colx >= 1 && colx <= a.g.uN - 2)
                  g = (float)(((double)Tv + (double)__fmul_rn(lambd, av[c])) + (double)__fsub_rn(uv, tv) / 4.0);
                else
                  g = __fadd_rn(__fmul_rn(lambd, av[c]), __fmul_rn(__fsub_rn(uv, tv), 0.5f));
                const uint32_t gb = __float_as_uint(g) & 0x7FFFFFFFu, ub = __float_as_uint(uv) & 0x7FFFFFFFu;
                mgb[c] = mgb[c] > gb ? mgb[c] : gb;
                mub[c] = mub[c] > ub ? mub[c] : ub;
                mu[c] = __builtin_fmaxf(mu[c], uv);
                gout[c] = __float_as_uint(g);
              }
              rflags |= 64u;
              if (pam) __builtin_amdgcn_raw_buffer_store_b96(gout, rs_o, voff, so, ICS_EPI_STORE_AUX);
            }
          }
        }
      }
      };
      auto run_all = [&](auto tloc) {
        if (MODE == 1 && a.tv_kind != 0) run_epi(std::integral_constant<int, 1>{}, std::true_type{}, tloc);
        else run_epi(std::integral_constant<int, ICS_EPI_TB(MODE)>{}, std::false_type{}, tloc);
      };
      if (NH == 1 || half == 0) run_all(std::integral_constant<int, 0>{}); else run_all(std::integral_constant<int, C::PAIR_SETS ? 1 : C::RS / 2>{});
    }
    ICS_TICK(5);
    // (the next tile's first barrier, after the per-wave maxima, also orders this tile's fragment reads
    //  before the next conversion overwrites the planes)
  }

  ICS_TICK(6);
  if (a.sched && tid == 0) {   // the last workgroup out re-arms the counters for the next launch
    if (atomicAdd(a.sched + 8, 1u) == gridDim.x - 1) {
#pragma unroll
      for (int i = 0; i < 9; ++i) __hip_atomic_store(a.sched + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (MODE == 1) {
    // step-size reductions of all tiles of this workgroup: wave shuffle -> LDS -> one atomic per value
    uint32_t kg[3], ku[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      kg[c] = mgb[c] > 0x7F800000u ? 0xFFC00000u : ((rflags & 64u) ? ics_f2key(__uint_as_float(mgb[c])) : 0u);   // NaN propagates like np.amax
      ku[c] = mub[c] > 0x7F800000u ? 0xFFC00000u : ((rflags & 64u) ? ics_f2key(mu[c]) : 0u);
      kg[c] = wave_max_u32(kg[c]); ku[c] = wave_max_u32(ku[c]);
    }
    uint32_t* red_lds = reinterpret_cast<uint32_t*>(fscr);
    if (lane == 0) {
#pragma unroll
      for (int c = 0; c < 3; ++c) { red_lds[wv * 8 + c] = kg[c]; red_lds[wv * 8 + 3 + c] = ku[c]; }
    }
    __syncthreads();
    // (lane index re-derived here: computed from `tid` the address of the lane's key was hoisted above the tile loop as a
    //  64-bit VGPR pair and spilled around it -- 8 bytes of scratch in the K = 9, 11, 29, 35, 37 back-projections)
    const int tq = opaque(tid);
    if (tq < 6) {
      uint32_t m = red_lds[tq];
#pragma unroll
      for (int w = 1; w < C::NW; ++w) { const uint32_t o2 = red_lds[w * 8 + tq]; m = m > o2 ? m : o2; }
      const int slot = tq < 3 ? ICS_RED_MAXG + tq : ICS_RED_MAXU + (tq - 3);
      if (m > a.red[slot]) atomicMax(a.red + slot, m);
    }
  }
  ICS_TICK(9);
  ICS_TICK_FLUSH;
}

template <int K, int MODE, int RS, int NH = 1>
hipError_t launch_one(const IcsConvArgs& a, hipStream_t s) {
  using C = MCfg<K, RS, NH>;
  static std::atomic<bool> configured[ICS_MAX_DEVICES];  // per device: the dynamic-LDS attribute is a per-device function property
  const int dev = ics_current_device();
  auto kern = k_conv_mfma<K, MODE, RS, NH>;
  if (hipError_t e = ics_configure_lds(configured, dev, kern, C::LDS_BYTES); e != hipSuccess) return e;
  const int ncu = ics_device_cus(dev);
  const int ntiles = MODE == 0 ? ((a.g.N + C::TW - 1) / C::TW) * ((a.g.M + C::TH - 1) / C::TH) : a.g.tiles_x * ((a.g.uM + C::TH - 1) / C::TH);
  int grid = C::WGS * ncu;                   // persistent workgroups: as many as fit the LDS of a CU
#ifdef ICS_GRID_WGS
  grid = ICS_GRID_WGS * ncu;
#endif
  if (grid > ntiles) grid = ntiles;
  // test hook: fewer persistent workgroups, so that small frames make every workgroup walk several tiles (next-tile
  // prefetch, band split) -- tests/test_gpu_rl.py::test_blind_golden_576x520_multi_tile_walk
  if (const int m = ics_debug().max_wgs.load(std::memory_order_relaxed); m > 0 && grid > m) grid = m;
  // Dynamic tile claiming (a.sched) pays when a workgroup walks many tiles -- the edge tiles are cheaper and the static walk leaves
  // a partial last round -- and costs when it walks few (the claim's round trip is exposed).  Measured on MI355X, static -> dynamic,
  // ms per inner iteration: 4096^2 15x15 (10.7 tiles per workgroup) blind 0.868 -> 0.847, non-blind 0.551 -> 0.545; 6144^2 31x31
  // (36) blind 4.45 -> 4.36; 3072^2 (6) level; 2560^2 (4.2) 0.385 -> 0.395; 2048^2 (2.7) non-blind 0.169 -> 0.180.
  // Hence: from 8 tiles per workgroup on.  ICS_DYNAMIC_TILES=0|1 forces either.
  IcsConvArgs b = a;
  if (b.sched) {
    const int dyn = ics_debug().dynamic_tiles.load(std::memory_order_relaxed);
    const bool on = dyn >= 0 ? dyn == 1 : ntiles >= 8 * grid;
    if (!on) b.sched = nullptr;
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(C::NT), C::LDS_BYTES, s, b);
  return hipGetLastError();
}

// Tile height per PSF size and frame, measured on MI355X (tools/bench_conv_mfma.hip, ms for mode 0 / mode 1, RS = 4 -> RS = 2):
//   4096^2: K = 9 0.139 / 0.183 -> 0.145 / 0.187, K = 13 level, K = 15 0.169 / 0.218 -> 0.167 / 0.207, K = 17 0.218 / 0.267 ->
//   0.184 / 0.218, K = 19 0.347 / 0.405 -> 0.325 / 0.364, K = 21 -5 % / -9 %, K = 23 .. 31 +10 .. 14 % (one workgroup per CU
//   either way, and the A-fragment reads then bound the step); 1024^2 .. 3072^2, K <= 15: RS = 2 ahead by 3 .. 30 % (finer
//   tiles balance the 256 CUs better).  Hence: K >= 23 -> 4; K = 15 .. 21 -> 2; K <= 13 -> 2 up to 3000 tiles of 64 x 64, else 4.
#ifndef ICS_MFMA_39_NH2
#define ICS_MFMA_39_NH2 1
#endif
template <int K> struct TileRs {
  static constexpr bool has2 = ICS_MFMA_ALL_RS || K <= 21 || K >= 39;   // 39 .. 49: the planes of a 64-row tile do not fit the LDS
  static constexpr bool has4 = ICS_MFMA_ALL_RS || K <= 13 || (K >= 23 && K <= 37);
  // 16-row tiles (fragment rows 1 apart, one accumulator set; round 4) for frames that do not give every CU a 32-row tile: the 255-px
  // windows of deblur_module's blind phase (45 tiles of 32 x 64 on 256 CUs) and 512^2 (128).  A fragment read then feeds 3 MFMAs only --
  // irrelevant where a kernel is one chain of dependent round trips per workgroup; what counts is that the chain is half as long.
  static constexpr bool has1 = !ICS_MFMA_NO_RS1 && K <= 15;
};
// does this frame take the 16-row tiles?  (fewer 32-row tiles than compute units)
template <int K>
static bool small_frame_rs1(int mode, const IcsGeom& g) {
  if (!TileRs<K>::has1) return false;
  const int frs = ics_debug().conv_rs.load(std::memory_order_relaxed);
  if (frs == 1) return true;
  if (frs == 2 || frs == 4) return false;
  const long t32 = mode == 0 ? (long)((g.N + 63) / 64) * ((g.M + 31) / 32) : (long)g.tiles_x * ((g.uM + 31) / 32);
  return t32 < (long)ics_device_cus(ics_current_device());
}
template <int K>
hipError_t launch_k(int mode, const IcsConvArgs& a, hipStream_t s) {
  if constexpr (TileRs<K>::has1) {
    if (small_frame_rs1<K>(mode, a.g)) return mode == 0 ? launch_one<K, 0, 1>(a, s) : launch_one<K, 1, 1>(a, s);
  }
  bool rs2 = K >= 39 || (K <= 21 && (K >= 15 || (long)a.g.tiles_x * a.g.tiles_y <= 3000));
  if (TileRs<K>::has2 && TileRs<K>::has4) {   // test / harness hook: force a tile height where both are built
    const int frs = ics_debug().conv_rs.load(std::memory_order_relaxed);
    rs2 = frs == 2 ? true : (frs == 4 ? false : rs2);
  }
  if constexpr (TileRs<K>::has2) {
    if constexpr (K >= 39 && ICS_MFMA_39_NH2) {   // 39 .. 49: 32-row tiles with the kernel rows split between two waves per column block
      if (rs2 || !TileRs<K>::has4) return mode == 0 ? launch_one<K, 0, 2, 2>(a, s) : launch_one<K, 1, 2, 2>(a, s);
    } else {
      if (rs2 || !TileRs<K>::has4) return mode == 0 ? launch_one<K, 0, 2>(a, s) : launch_one<K, 1, 2>(a, s);
    }
  }
  if constexpr (TileRs<K>::has4) {
    if constexpr (K >= 23) {
      // one workgroup per CU: 8 waves, kernel rows split between the two waves of a column block (MCfg NH = 2).  Measured against the
      // 4-wave form (ICS_TEST_CONV_NH=1 in a build with ICS_MFMA_ALL_RS): 4096^2 K = 23 0.417 / 0.485 -> 0.390 / 0.448 ms,
      // 6144^2 K = 31 1.089 / 1.266 -> 1.077 / 1.197 ms
      const int fnh = ICS_MFMA_ALL_RS ? ics_debug().conv_nh.load(std::memory_order_relaxed) : 0;
      if (fnh != 1) return mode == 0 ? launch_one<K, 0, 4, 2>(a, s) : launch_one<K, 1, 4, 2>(a, s);
    }
    if constexpr (K < 23 || ICS_MFMA_ALL_RS) return mode == 0 ? launch_one<K, 0, 4>(a, s) : launch_one<K, 1, 4>(a, s);
  }
  return hipErrorInvalidValue;
}

}  // namespace

// Translation units: the 36 kernel instances take minutes to compile, so libics_hip.so builds them in three parts
// (ICS_MFMA_PART = 0: K <= 17 + the public entry points, 1: K = 19..27, 2: K = 29..37, 3: K = 39..49; ics_conv_mfma_p1/_p2/_p3.hip include this
// file).  Without ICS_MFMA_PART (tools/) everything is in one unit.
#ifndef ICS_MFMA_PART
#define ICS_MFMA_ALL 1
#define ICS_MFMA_PART 0
#else
#define ICS_MFMA_ALL 0
#endif

hipError_t ics_launch_conv_mfma_part1(int mode, const IcsConvArgs& a, hipStream_t s);
hipError_t ics_launch_conv_mfma_part2(int mode, const IcsConvArgs& a, hipStream_t s);
hipError_t ics_launch_conv_mfma_part3(int mode, const IcsConvArgs& a, hipStream_t s);

#if ICS_MFMA_ALL || ICS_MFMA_PART == 3
// 39 .. 49 (round 3): two 32-wide windows still cover the 16 + K - 1 input columns of a column block, so the cost per kernel row is
// that of 19 .. 37; 32-row tiles, four waves, one workgroup per CU (the six planes + the weight rows take 112 .. 147 KB)
hipError_t ics_launch_conv_mfma_part3(int mode, const IcsConvArgs& a, hipStream_t s) {
  switch (a.g.K) {
    case 39: return launch_k<39>(mode, a, s);
    case 41: return launch_k<41>(mode, a, s);
    case 43: return launch_k<43>(mode, a, s);
    case 45: return launch_k<45>(mode, a, s);
    case 47: return launch_k<47>(mode, a, s);
    case 49: return launch_k<49>(mode, a, s);
    default: return hipErrorInvalidValue;
  }
}
#endif
#if ICS_MFMA_ALL || ICS_MFMA_PART == 1
hipError_t ics_launch_conv_mfma_part1(int mode, const IcsConvArgs& a, hipStream_t s) {
  switch (a.g.K) {
    case 19: return launch_k<19>(mode, a, s);
    case 21: return launch_k<21>(mode, a, s);
    case 23: return launch_k<23>(mode, a, s);
    case 25: return launch_k<25>(mode, a, s);
    case 27: return launch_k<27>(mode, a, s);
    default: return hipErrorInvalidValue;
  }
}
#endif
#if ICS_MFMA_ALL || ICS_MFMA_PART == 2
hipError_t ics_launch_conv_mfma_part2(int mode, const IcsConvArgs& a, hipStream_t s) {
  switch (a.g.K) {
    case 29: return launch_k<29>(mode, a, s);
    case 31: return launch_k<31>(mode, a, s);
    case 33: return launch_k<33>(mode, a, s);
    case 35: return launch_k<35>(mode, a, s);
    case 37: return launch_k<37>(mode, a, s);
    default: return hipErrorInvalidValue;
  }
}
#endif

#if ICS_MFMA_PART == 0
bool ics_conv_mfma_supported(int K) { return K >= 3 && K <= 49 && (K & 1); }   // 16 + K - 1 <= 64: two MFMA windows per column block

// Measured on MI355X at 4096^2: ahead of the packed-fp32 kernels at every size built (K = 19, 21 were level with
// 64-row tiles -- two 32-wide windows per column block, one workgroup per CU -- and are ~8 % ahead with 32-row tiles).
bool ics_conv_mfma_preferred(int K) { return ics_conv_mfma_supported(K); }

// Tile height (fragment row stride RS = 2: 32 rows, 4: 64 rows) launch_k() picks for this PSF size and frame, 0 for the 8-wave
// kernels (K >= 23): the caller prepares the accumulator-order image (ics_image_acc.h) of that layout for mode 0.
int ics_conv_mfma_rs(int K, const IcsGeom& g, int cus) {   // cus < 0: the current device's (launch paths); the shape-only describe path passes a count
  if (K >= 23) return 0;
  if (K <= 15) {   // 16-row tiles on small frames (TileRs::has1): no accumulator-order image for them
    const int frs = ics_debug().conv_rs.load(std::memory_order_relaxed);
    const long t32 = (long)((g.N + 63) / 64) * ((g.M + 31) / 32);
    if (!ICS_MFMA_NO_RS1 && (frs == 1 || (frs == 0 && t32 < (long)(cus >= 0 ? cus : ics_device_cus(ics_current_device()))))) return 0;
  }
  bool rs2 = K <= 21 && (K >= 15 || (long)g.tiles_x * g.tiles_y <= 3000);
  if (K <= 13) {   // both heights are built
    const int frs = ics_debug().conv_rs.load(std::memory_order_relaxed);
    rs2 = frs == 2 ? true : (frs == 4 ? false : rs2);
  }
  return rs2 ? 2 : 4;
}

// weight table: [c][a] rows of 2 * WROWB bytes, hi/lo dword-interleaved (the LDS image), then one float 1/s_w (ics_common.h)
size_t ics_conv_mfma_table_floats(int K) { return (size_t)3 * K * 2 * (((2 * (K + 17) + 3) & ~3) / 4) + 4; }

hipError_t ics_launch_conv_mfma(int mode, const IcsConvArgs& a, hipStream_t s) {
  if ((mode != 0 && mode != 1) || !a.bt) return hipErrorInvalidValue;
#ifdef ICS_MFMA_ONLY_K   /* experiments: one PSF size per build (scripts/isa_one.sh) */
  return a.g.K == ICS_MFMA_ONLY_K ? launch_k<ICS_MFMA_ONLY_K>(mode, a, s) : hipErrorInvalidValue;
#else
  switch (a.g.K) {
    case 3: return launch_k<3>(mode, a, s);
    case 5: return launch_k<5>(mode, a, s);
    case 7: return launch_k<7>(mode, a, s);
    case 9: return launch_k<9>(mode, a, s);
    case 11: return launch_k<11>(mode, a, s);
    case 13: return launch_k<13>(mode, a, s);
    case 15: return launch_k<15>(mode, a, s);
    case 17: return launch_k<17>(mode, a, s);
    default: return a.g.K <= 27 ? ics_launch_conv_mfma_part1(mode, a, s) : (a.g.K <= 37 ? ics_launch_conv_mfma_part2(mode, a, s) : ics_launch_conv_mfma_part3(mode, a, s));
  }
#endif
}
#endif
