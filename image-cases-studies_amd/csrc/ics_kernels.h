// ics_kernels.h -- launchers of the non-convolution kernels (ics_kernels.hip, ics_stats.hip, ics_filters.hip).
#pragma once
#include "ics_common.h"

// ---- A5/A6/A8/A10 (lib/deconvolution.pyx:499-552): image update + DoF blend -------------------
struct IcsUpdateArgs {
  const float* u;      // frame origin of the current u
  float* u_out;        // frame that receives the updated u (== u for an in-place update)
  const float* ut;     // majoriser (pyx:462)
  const float* g;      // raw back-projection (A3)
  const float* f;      // image
  const float* tv;     // extended modes: T frame (NULL in the shipped mode)
  int tv_kind;         // 0 shipped, 1 active MM-TV, 2/3 PAM
  float* f_rw;         // active MM-TV: the image frame is updated in place (pyx:549)
  const uint32_t* red; // reduction keys of this inner iteration (ICS_RED_*)
  float* scal;         // device scalar block (ICS_SC_*): dt, maxu, maxg are recorded
  uint32_t* dofkeys;   // [0] = min key, [1] = max key, [2] = NaN flag (only when want_dof)
  float step, lambd;
  int blind;
  int want_dof;
  IcsGeom geo;
};
hipError_t ics_launch_update(const IcsUpdateArgs& a, hipStream_t s);

// ---- active MM-TV (build-defined extension, pyx:517/:543 made reachable): TV term of u against ut ---
struct IcsTvTermArgs {
  const float* u;      // frame origin
  const float* ut;     // majoriser
  const float* f;      // image (for max image_k, pyx:548)
  float* tv;           // T frame written
  uint32_t* red;       // ICS_RED_MAXT / ICS_RED_MAXF keys of this inner iteration
  float epsilon;       // 1e-2 blind / 1e-6 non-blind (pyx:434-437)
  int kind;            // 1 MM-TV term, 2 isotropic TV gradient, 3 collaborative L-inf,1,1 TV gradient
  IcsGeom geo;
  int planar = 0;      // kinds 2 / 3 only: u and tv are ORIGINS of channel-planar mirrors (ics_common.h), the transform-tile pipeline's frames
};
hipError_t ics_launch_tvterm(const IcsTvTermArgs& a, hipStream_t s);

// ---- A13 (pyx:567-571): PSF gradient, fp32 MFMA ------------------------------------------------
struct IcsGradkArgs {
  const float* e;   // residual frame origin (zero outside the M x N interior)
  const float* u;   // u frame origin
  float* partial;   // [nblocks][3][16nb][16nb] per-workgroup partial sums
  IcsGeom geo;
  int planar;       // matrix-core kernel only: e and u are ORIGINS of channel-planar mirrors (ics_common.h, the FFT-tile pipeline)
};
int ics_gradk_blocks(const IcsGeom& g, int cus);  // grid size (persistent workgroups)
hipError_t ics_launch_gradk(const IcsGradkArgs& a, int nblocks, hipStream_t s);
// matrix-core variant for PSF sizes <= 15 (ics_gradk_mfma.hip): fp16-split operands, same partial layout
bool ics_gradk_mfma_supported(int K);
hipError_t ics_launch_gradk_mfma(const IcsGradkArgs& a, int nblocks, hipStream_t s);
// ---- A11 + A13 fused (ics_synth_gradk_mfma.hip, PSF sizes <= 15): e' = conv(u, psf) - image never leaves the CU ---------
struct IcsFusedArgs {
  const float* u;     // u frame origin
  const float* f;     // image frame origin
  float* e_out;       // residual frame origin: e' is stored only where tiles meet the window below (or everywhere: store_all)
  const void* bt;     // weight table of the matrix-core convolution, conv orientation (ics_common.h)
  float* partial;     // [nblocks][3][16][16] per-workgroup partial sums (layout of k_gradk_mfma, reduced by k_gradk_reduce)
  int wy0, wy1, wx0, wx1;   // stats window (pyx:600-601,627) in u-frame coordinates
  int store_all;
  const float* facc;  // the image in accumulator order (ics_image_acc.h) in the layout of `rs`, or NULL: the epilogue reads the HWC frame
  int rs;             // tile height: 4 = 64-row tiles, two workgroups per CU (k_synth_gradk); 2 = 32-row tiles, three per CU (k_synth_gradk2)
  IcsGeom g;
};
bool ics_synth_gradk_supported(int K);
hipError_t ics_launch_synth_gradk(const IcsFusedArgs& a, int nblocks, hipStream_t s);
// gradk[a][b][c] = sum over workgroups (double accumulation, fixed order)
hipError_t ics_launch_gradk_reduce(const float* partial, int nblocks, float* gradk, const IcsGeom& g, hipStream_t s);
// one La x Lb tap block (partial blocks of 3 * nt * nt floats) into rows a0.., columns b0.. of a Kf x Kf gradient
hipError_t ics_launch_gradk_reduce_block(const float* partial, int nblocks, float* gradk, int nt, int La, int Lb, int Kf, int a0, int b0, hipStream_t s);

// ---- row bands: see include/ics_hip.h ICS_STAGE_BAND_* ----------------------------------------------
hipError_t ics_launch_band_reduce(const float* gr, const float* u, const float* ut, const IcsGeom& g, float lambd, int r0, int r1, uint32_t* red, hipStream_t s);
hipError_t ics_launch_band_mask_e(float* e, const IcsGeom& g, int i0, int i1, hipStream_t s);

// ---- channel-planar mirrors (ics_planar.hip) and the FFT-tile convolution (ics_conv_fft.hip) --------------------------------------------
// src / dst: buffer STARTS; rows [y0, y1), pixels [x0, x1) in u-frame coordinates (widened to 4-pixel groups), or the whole buffer
hipError_t ics_launch_planar_convert(bool to_planar, const float* src, float* dst, const IcsGeom& g, bool whole, int y0, int y1, int x0, int x1, hipStream_t s);
hipError_t ics_launch_update_planar(const IcsUpdateArgs& a, hipStream_t s);   // frame pointers = origins of planar mirrors
bool ics_conv_fft_supported(int K);
size_t ics_conv_fft_spectrum_floats();                                        // per orientation
// (blk_n > 0: the spectra of blk_n x blk_n tap blocks of blk_k x blk_k taps, block q at spec + q * ics_conv_fft_spectrum_floats())
hipError_t ics_launch_fft_spectrum(const float* psf, int K, float* spec_conv, float* spec_corr, hipStream_t s, int blk_n = 0, int blk_k = 0);
// PSF sizes above the single-tile range (99 ... 255): the convolutions and the PSF gradient as tap blocks on the tiles -- the blocks' products
// are summed in the frequency domain, one inverse transform per unit (k_conv_fft_blk); modes 0 and 1 of the shipped loop
bool ics_conv_fft_blk_supported(int K);
void ics_conv_fft_blk_shape(int K, int* blk_n, int* blk_k);
hipError_t ics_launch_conv_fft_blk(int mode, const IcsConvArgs& c, const float* spec, int blk_n, int blk_k, hipStream_t s);
hipError_t ics_launch_gradk_fft_blk(const float* u, const float* e, const IcsGeom& g, int blk_n, int blk_k, float* partial, float* gradk, hipStream_t s);
// modes 0 and 1 of ics_launch_conv; `planar` = bit mask of the frames of `a` that are origins of planar mirrors (ICS_FFT_PL_*)
#define ICS_FFT_PL_IN 1
#define ICS_FFT_PL_OUT 2
#define ICS_FFT_PL_F 4
#define ICS_FFT_PL_U 8
#define ICS_FFT_PL_UT 16
#define ICS_FFT_PL_TV 32
#define ICS_FFT_PL_ALL 63
hipError_t ics_launch_conv_fft(int mode, const IcsConvArgs& a, const float* spec, int planar, hipStream_t s);
// mode 2 (k_conv_fft<2>): A1 + A2 + A3 in ONE unit per tile pair -- interior tiles stay in the frequency domain between the two convolutions
// (one forward, one inverse transform; the image enters as the precomputed spectra of its windows), the tiles of the outer ring mask the
// residual in between.  Valid output 128 - 2 K + 2 pixels a side: for small PSFs.  The residual frame is NOT written.
hipError_t ics_launch_conv_fft_region(const IcsConvArgs& c, const float* spec, int oy0, int ox0, int oy1, int ox1, hipStream_t s);   // mode 0 over the tiles of a window
size_t ics_conv2_fft_fspec_floats(const IcsGeom& g);
bool ics_conv2_fft_supported(const IcsGeom& g);
hipError_t ics_launch_fft_image_spectrum(const float* f, const IcsGeom& g, float* fspec, hipStream_t s);
hipError_t ics_launch_conv2_fft(const IcsConvArgs& c, const float* spec_conv, const float* spec_corr, const float* fspec, hipStream_t s);
// A12 + A13 on the same tiles (fp32): u, e = origins of planar mirrors; partial = ics_gradk_fft_blocks(cus) * K * K floats of scratch
int ics_gradk_fft_blocks(int cus);
hipError_t ics_launch_gradk_fft(const float* u, const float* e, const IcsGeom& g, float* partial, float* gradk, hipStream_t s);
// A11 + A12 + A13 as one unit per tile pair (three transforms instead of four; the residual stays in the tile buffer and is stored only for
// tiles under the window [wy0, wy1) x [wx0, wx1) of u-frame coordinates, or everywhere with store_all): bit-identical to
// ics_launch_conv_fft(0) followed by ics_launch_gradk_fft
hipError_t ics_launch_synth_gradk_fft(const float* u, const float* f, float* e, const float* spec, const IcsGeom& g, int wy0, int wy1, int wx0, int wx1, int store_all,
                                      float* partial, float* gradk, hipStream_t s);

// ---- zero-fill of up to ICS_ZERO_MAX device blocks in one launch (the ~24 buffers of a new job: one launch instead of 24 memsets) ----
#define ICS_ZERO_MAX 32
struct IcsZeroArgs {
  void* p[ICS_ZERO_MAX];                 // 16-byte aligned
  unsigned long long end16[ICS_ZERO_MAX]; // running total of 16-byte units up to and including block i
  int count;
};
hipError_t ics_launch_zero_many(const IcsZeroArgs& a, hipStream_t s);

// the small device state of a job at the start of ics_rl_run, in one launch (was five memsets and two host -> device copies)
struct IcsRunResetArgs {
  int* flags;          // [4]  := 0
  uint32_t* sched;     // [16] := 0
  double* dacc;        // [8]  := 0
  uint32_t* ukey;      // [2]  := 0
  uint32_t* red;       // [nred] := 0
  int nred;
  uint32_t* dofkeys;   // [8] := {0xFFFFFFFF, 0, 0, 0} x 2
};
hipError_t ics_launch_run_reset(const IcsRunResetArgs& a, hipStream_t s);

// ---- A14-A17 (pyx:574-589) + weight packing ---------------------------------------------------
struct IcsPsfArgs {
  float* psf;          // [K][K][3] local psf (pyx: the name `psf` inside the function)
  const float* gradk;  // [K][K][3]
  float* wconv;        // [K+1][wrow] row-pair packed rot180(psf): weights of A1 (correlation orientation)
  float* wcorr;        // [K+1][wrow] row-pair packed psf:         weights of A3
  void* bt_conv;       // Toeplitz fragment tables of the matrix-core convolution (ics_common.h), or NULL
  void* bt_corr;
  float* psf_caller;   // what the caller's array holds (correlation quirk, pyx:585)
  float* work;         // PSF sizes above 63 only: 3*K*K floats of scratch (the kernel's working copy; LDS below that)
  float* scal;         // ICS_SC_DTPSF recorded
  int* frozen;         // device flag: caller array detached (pyx:585 rebinding)
  float step;
  int K, wrow;
  int correlation;
  int do_step;         // 0: only (re)pack the weights from psf
};
hipError_t ics_launch_psf(const IcsPsfArgs& a, hipStream_t s);
// PSF sizes 65 ... 127 (ics_big.hip): run-time-sized fp32 kernels; the convolution reads the PSF itself (rot180 for mode 0)
bool ics_big_supported(int K);
hipError_t ics_launch_conv_big(int mode, const IcsConvArgs& a, const float* psf, hipStream_t s);
hipError_t ics_launch_gradk_big(const IcsGradkArgs& a, int nblocks, hipStream_t s);
// tap blocks on the matrix cores (PSF sizes 51 ... 127, ics_big.hip): nblk x nblk weight tables of Kb x Kb taps, and the frame sum
hipError_t ics_launch_pack_blocks(const float* psf, int K, int Kb, int nblk, void* tconv, void* tcorr, size_t table_floats, hipStream_t s);
hipError_t ics_launch_frame_add(float* out, const float* add, int pitch, int y0, int y1, int f0, int f1, hipStream_t s);
hipError_t ics_launch_frame_neg(float* out, const float* in, size_t count, hipStream_t s);   // out = -in over a whole frame buffer

// ---- the inner iterations of an outer iteration as one cooperative launch, small frames (ics_small.hip) ------------------------------
// threads of a workgroup and outputs per thread and kernel row.  (1024 threads with 8 outputs -- all that fits 128 registers -- measured slower at
// every PSF size it was tried for, 3 ... 15: 255^2 / 15 blind 0.051 -> 0.063 ms per inner iteration; more LDS reads and partial sums per product)
constexpr int ics_small_threads(int K) { return 512; }
constexpr int ics_small_cw(int K) { return 16; }
#define ICS_SMALL_MAX_K 31
#define ICS_SMALL_LDS_BYTES 163840
#define ICS_SMALL_BAR_GROUPS 16
#define ICS_SMALL_BAR_WORDS (16 * ICS_SMALL_BAR_GROUPS)   /* 64-bit counters, one per 128 bytes: [16 g] = arrivals of the workgroups w with w % 16 == g */
struct IcsSmallPlan {
  int T, tiles_x, tiles_y, nwg;       // tile edge (32 / 64), tiles of the u-frame, workgroups = 3 x tiles
  int HU, pU, H1, pE, pT;             // LDS regions: U (T + 4 pad)^2 at pitch pU, E and F (T + 2 pad)^2 at pE, UT and G T^2 at pT (odd pitches)
  int chunks1, Cp1, AG1, AGn1;        // A1 on the tile +- pad: 16-column chunks, pitch of the partial sums, kernel rows per group, groups
  int chunksT, CpT, AGT, AGnT;        // A3 / A11 on the tile
  int YG, rpy;                        // A13: row groups of rpy rows
  int off_U, off_E, off_F, off_UT, off_G, off_P, off_W, off_PSF, lds_floats;
  uint32_t m_HU, m_H1, m_T, m_per1, m_perT, m_perG, m_YG;   // 2^32 / d + 1 of the divisors of the index arithmetic (ics_small.hip, udiv)
};
struct IcsSmallArgs {
  const float* u_in;        // u at the start of the outer iteration = its majoriser ut (pyx:462); read only
  float* u_out;             // receives u after every inner iteration
  const float* f;           // image
  float* e;                 // residual: the tile interiors of the last inner iteration (the stop test's operand)
  uint32_t* red;            // inner x ICS_RED_STRIDE keys: the maxima of every inner iteration are left here
  unsigned long long* keys; // inner x workgroups: every tile's max |gradu| (high word) and max u as order-preserving keys
  uint32_t* dofkeys;
  float* scal;
  float* psf; float* psf_caller; int* frozen;   // as IcsPsfArgs
  float* psf_bak;           // not NULL: psf and psf_caller as they are when the launch starts are copied here first (2 x 3 K^2)
  float* part;              // 3 x tiles x K^2: the tiles' shares of the PSF gradient
  float* gradk;             // [K][K][3]
  unsigned long long* bar;  // ICS_SMALL_BAR_WORDS counters, zeroed with the job's state
  unsigned long long bar_gen;   // barriers passed since then
  unsigned long long* trace;    // debug switch small_trace: 64 wall-clock stamps per workgroup, else NULL
  float step, lambd;
  int blind, correlation, inner;
  IcsSmallPlan plan;
  IcsGeom g;
};
// (64-pixel tiles -- frames up to ~570^2 -- are built but measured behind the multi-launch path: allow64 = debug switch small_iter = 2)
bool ics_small_plan(const IcsGeom& g, int cus, IcsSmallPlan* out, bool allow64 = false);
static inline int ics_small_barriers(int blind, int inner) { return blind ? 4 * inner : 2 * inner - 1; }
hipError_t ics_launch_small_iter(const IcsSmallArgs& a, hipStream_t s);

// ---- A18/A19 (pyx:593-638): window statistics and residual-whiteness metric -------------------
struct IcsStatsArgs {
  const float* e;      // residual frame origin
  const float* u;      // u frame origin
  float* scal;         // ICS_SC_MR / HU / VARU written
  uint32_t* dofkeys;   // DoF keys of the outer iteration (read; re-armed when `rearm`)
  uint32_t* red;       // reduction slots of the outer iteration's five inner iterations (8 * ICS_RED_STRIDE words), or nullptr
  int rearm;           // ics_rl_run: the kernel that writes the scalars also resets dofkeys and red for the next outer iteration
  double* dacc;        // 8 double accumulators (zeroed by the launcher)
  uint32_t* ukey;      // 2 keys (max |t|)
  float2* z;           // [3][P][P] complex scratch
  const float2* tw;    // [P/2] twiddles exp(-2 pi i k / P)
  const float* weights;// [H][W] Gaussian window (pyx:393-404)
  int top, bottom, left, right;
  int P, logP;
  int do_mr;
  IcsGeom geo;
};
hipError_t ics_launch_stats(const IcsStatsArgs& a, hipStream_t s);
hipError_t ics_launch_hasnan(const float* u, const IcsGeom& g, int* flag, hipStream_t s);

// ---- standalone operators ----------------------------------------------------------------------
hipError_t ics_launch_tv(const float* u, int M, int N, float eps, int order, int norm, float* out, float* div, hipStream_t s);
// one LDS-tiled pass: out = conv2d_symm(src, kern); usm: out = src0 + (src0 - conv) * amount (lib/utils.py:275)
hipError_t ics_launch_conv2d_symm(const double* src, int H, int W, const double* kern, int KH, int KW, double* out,
                                  const double* src0, int usm, double amount, hipStream_t s);
// ws: (2 radius + 1)^2 spatial weights exp(-(i^2 + j^2) / (2 std_s^2)), j slow (the reference's offset order)
hipError_t ics_launch_bilateral(const double* src, int H, int W, int radius, double std_i, const double* ws, double* out, hipStream_t s);

// ---- bicubic resize between pyramid levels (ics_resize.hip; reference deconvolve.py:245-249) -------------
size_t ics_resize_scratch_doubles(int H, int W, int C);
hipError_t ics_launch_resize(double* src, int H, int W, int C, const double* wy, int ry, const double* wx, int rx, double* scratch,
                             double* out, int OH, int OW, hipStream_t s);
hipError_t ics_launch_resize_f32(const float* src, int H, int W, int C, const double* wy, int ry, const double* wx, int rx, double* scratch,
                                 float* out, int OH, int OW, hipStream_t s);   // float32 in / out, float64 inside

// ---- device-resident images (ics_img.hip; deconvolve.py:24-37, :100-103, :346-352) ----------------------
hipError_t ics_launch_img_pad_edge(const float* in, int H, int W, float* out, int top, int bottom, int left, int right, hipStream_t s);
hipError_t ics_launch_img_gamma(float* a, long n, float div, float exponent, float mul, int clip01, hipStream_t s);
hipError_t ics_launch_f32_to_f64(const float* in, double* out, long n, hipStream_t s);
hipError_t ics_launch_int_to_f32(const void* in, int bytes_per_value, float* out, long n, hipStream_t s);   // uint8 / uint16 -> float32
hipError_t ics_launch_f64_to_f32(const double* in, float* out, long n, hipStream_t s);
