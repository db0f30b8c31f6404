/*
 * ics_hip.h -- C ABI of libics_hip.so: the MI355X (gfx950) implementation of the
 * Richardson-Lucy / MM deconvolution hot path of aurelienpierre/Image-Cases-Studies.
 *
 * This is the drop-in boundary.  Everything is `extern "C"`, plain pointers and sizes; host
 * arrays are float32, C-contiguous, HWC (channel fastest, 3 channels) exactly as the reference
 * passes numpy buffers to lib/deconvolution.pyx.  Every entry point returns 0 on success or a
 * negative ICS_E* code; ics_last_error() returns a thread-local description (HIP error string
 * included).  No entry point falls back to a CPU path: without a usable gfx950 device every
 * compute call fails with ICS_ENODEV.
 *
 * Reference interfaces replaced (file:line in /root/reference):
 *   ics_rl_*               lib/deconvolution.pyx:341-675   richardson_lucy_MM (cpdef, :341-342)
 *   ics_normalize_kernel   lib/deconvolution.pyx:47-75     normalize_kernel (cpdef, :73-75)
 *   ics_tv                 lib/deconvolution.pyx:137-239   TV (cdef)
 *   ics_conv2d_symm        lib/utils.py:237-264            bessel_blur / gaussian_blur
 *                                                          (scipy.signal.convolve2d same/symm)
 *   ics_usm                lib/utils.py:267-277            USM
 *   ics_bilateral          lib/utils.py:173-234            bilateral_filter
 *   ics_img_*              deconvolve.py:24-37,:94-103,:245-257,:303,:322-323,:346-352
 *                                                          the frames the driver keeps between two
 *                                                          richardson_lucy_MM calls (pad_image, gamma, window
 *                                                          views, resize), device-resident
 *   ics_group_*            (none: the reference is single-process; SURVEY.md 8e defines image-per-GPU sharding)
 *   ics_resize_bicubic     deconvolve.py:245-249           skimage.transform.resize(order=3, mode="edge")
 *                                                          between pyramid levels (un-vendored dependency of
 *                                                          the reference: restated on scipy.ndimage semantics)
 * The Python side that binds these (ctypes) is image-cases-studies_amd/lib/_native.py; the
 * reference-side binding a maintainer would add is shown in INTEGRATION.md.
 */
#ifndef ICS_HIP_H
#define ICS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ICS_ABI_VERSION 4 /* 3: self-describing ics_rl_params / ics_rl_stats (struct_size first), caller-owned traces,
                             per-outer-iteration callback, ics_group_allreduce_sum
                             4: the callback returns int (non-zero = stop after this outer iteration, stats.stopped = 2);
                                ics_rl_describe (which kernel family a run will use); the DoF ratio at exact 0/0 is 1 */

/* error codes */
#define ICS_OK 0
#define ICS_EINVAL (-1)  /* bad argument                                        */
#define ICS_ENODEV (-2)  /* no usable HIP device / wrong architecture            */
#define ICS_EHIP (-3)    /* a HIP runtime call failed (see ics_last_error)       */
#define ICS_ENOMEM (-4)  /* device or host allocation failed                     */
#define ICS_ESTATE (-5)  /* call sequence error (e.g. run before upload)         */
#define ICS_ENOSUP (-6)  /* unsupported size: PSF > 255, stats window > 4096 px, frame > 2 GiB */

typedef struct ics_ctx ics_ctx; /* one per (process, device): stream, events, scratch   */
typedef struct ics_rl ics_rl;   /* one deconvolution job: device-resident frames        */

/* ---- library / device ------------------------------------------------------------------ */
int ics_abi_version(void);
const char *ics_last_error(void);
int ics_device_count(int *count);
/* Creates a context on `device` (hipSetDevice + one non-blocking stream).  Fails with
 * ICS_ENODEV when the device is not gfx950. */
int ics_ctx_create(int device, ics_ctx **out);
void ics_ctx_destroy(ics_ctx *ctx);
int ics_ctx_synchronize(ics_ctx *ctx);
/* Device time in ms of the kernels of the last ics_conv2d_symm / ics_usm / ics_bilateral call (HIP events, transfers excluded). */
int ics_ctx_last_kernel_ms(ics_ctx *ctx, float *ms);
/* Device description: name (<=255 chars), compute units, HBM bytes. */
int ics_ctx_info(ics_ctx *ctx, char *name, size_t name_len, int *compute_units, uint64_t *hbm_bytes);

/* ---- Richardson-Lucy / MM job (lib/deconvolution.pyx:341-675) --------------------------- */

/* Called by ics_rl_run on the calling thread right after each outer iteration's statistics are known (pyx:593-659: the point
 * where the reference prints its progress lines), so that a binding can print them live.  `it` = outer iterations completed
 * (1-based), `stopped` = the stop test fired on this iteration.
 * Return value: 0 = go on; non-zero = ABORT: ics_rl_run leaves the loop after this outer iteration exactly as if `iterations`
 * had been `it` (u, psf and the statistics are those of iteration `it`, downloadable as usual) and reports
 * ics_rl_stats.stopped = 2.  This is the channel for deconvolve.py:338-342, which swallows a KeyboardInterrupt and keeps the
 * partial, in-place-updated `u`: a binding turns the interrupt it catches inside the callback into a non-zero return. */
typedef int (*ics_rl_progress_fn)(void *user, int it, int stopped, float dof_min, float dof_max, float M_r, float Hu, float varu);

/* Arguments of richardson_lucy_MM that are scalars (pyx:341-342).  `p, norm, order, priority,
 * refocus` are accepted and ignored by the reference and therefore absent.
 * struct_size MUST be sizeof(ics_rl_params) of the header the caller was built against: the library refuses any other
 * value with ICS_EINVAL instead of reading fields the caller never wrote (ics_rl_params_size() returns what it expects). */
typedef struct ics_rl_params {
  uint32_t struct_size;         /* = sizeof(ics_rl_params)                                 */
  int top, bottom, left, right; /* stats window, image coordinates (pyx:600-601,627)      */
  float tau;                    /* non-blind stop threshold (pyx:652)                      */
  int iterations;               /* max OUTER iterations, 5 inner each (pyx:375,460)        */
  float step_factor;            /* pyx:524,574                                             */
  float lambd;                  /* pyx:519                                                 */
  int blind;                    /* pyx:434,501,555                                         */
  int correlation;              /* pyx:584-585 (channel tie + caller-array rebinding quirk)*/
  int channels;                 /* `C`: bounds the blind channel loops (pyx:557,570); 3    */
  int tv_mode;                  /* ICS_TV_*: 0 = shipped behaviour (TV term dead)          */
  int stop_test;                /* 1 = evaluate the residual-whiteness stop test (pyx:623-654)
                                   on device every outer iteration (the reference always
                                   does); 0 = never stop early, M_r not computed;
                                   2 = compute M_r every outer iteration but never stop
                                   (fixed-length benchmark runs with the full workload)    */
  int profile;                  /* k > 0: bracket the kernel launches of every k-th inner iteration (k = 1:
                                   all) and the per-outer kernels with HIP events on the job's stream;
                                   per-kernel averages are reported in ics_rl_stats         */
  int fuse;                     /* 1: the image update of inner iteration i is fused in front of the next
                                   convolution (one kernel, u ping-pong, bit-identical results).  Default 0:
                                   measured SLOWER on MI355X at 4096^2/15x15 (0.67 ms vs 0.28 + 0.19 ms),
                                   the 1.8x halo recompute with two IEEE divisions per element outweighs
                                   the saved frame pass (NOTES_r01.md section 4)                            */
  int conv;                     /* ICS_CONV_*: which kernels run the convolutions A1/A3 and the PSF gradient A13    */
  int flags;                    /* ICS_FLAG_* bits, 0 = defaults                                                     */
  int band_row0, band_row1;     /* ICS_STAGE_BAND_* only: the rows [row0, row1) this job OWNS when one image is split into
                                   row bands over several jobs / GPUs; u-frame rows for BAND_REDUCE,
                                   image rows for BAND_MASK_E.  Ignored everywhere else.                                */
  ics_rl_progress_fn progress;  /* NULL or the per-outer-iteration callback above (ics_rl_run only)                   */
  void *progress_user;
} ics_rl_params;
size_t ics_rl_params_size(void); /* sizeof(ics_rl_params) as the library was built */

#define ICS_FLAG_NO_FUSED_GRADK 1 /* blind, matrix-core path, MK <= 15: run A11 and A13 as two kernels (k_conv_mfma<K,0> +
                                     k_gradk_mfma) instead of the fused k_synth_gradk (ics_synth_gradk_mfma.hip); env
                                     ICS_FUSED_GRADK=0 does the same for an unmodified caller.  With the fused kernel the
                                     residual frame (ICS_BUF_ERROR) holds e' of pyx:555-565 only on the 64x64 tiles that
                                     meet the stats window -- the only place the loop reads it (pyx:600-601,627)         */
#define ICS_FLAG_STAGE_ASYNC 2    /* ics_rl_stage: return when the stage is queued on the job's stream instead of when it has run.  The ordering
                                     contract, as implemented: ics_rl_read*, ics_rl_scalars, ics_rl_copy_rows and the all-reduces are queued on
                                     (or synchronise with) the job's stream, so they see every stage queued before them.  ics_rl_exchange_rows
                                     is HOST-SYNCHRONOUS on both sides: it first drains the job's stream, runs its send / receive on the
                                     group's own stream and returns when that stream has drained -- stages queued after it see the halo rows,
                                     stages queued before it have finished.  A stage under ICS_CONV_FFT converts its frames back and drains the
                                     stream whatever this flag says.  So a sequence of stages needs no host synchronisation of its own
                                     (lib/banded.py BandRank: one per outer iteration, where the stop decision is taken); code that removes the
                                     exchange's host waits must replace them by events between the two streams.  Ignored by ics_rl_run.  */

#define ICS_CONV_AUTO 0   /* the library's choice (ics_rl_describe tells): inside ics_rl_run the fp32 transform tiles on large frames -- from 0.5 ... 12 Mpx
                             depending on the PSF size, every size 3 ... 255 (csrc/ics_api.hip fft_preferred) --, the cooperative fp32
                             iteration kernel on small ones -- up to ~290 px a side, MK <= 31, shipped loop (ics_small.hip) -- and the fp16x2-split
                             matrix-core kernels between (convolutions: MK <= 49 directly, above as tap blocks of <= 33 x 33; PSF gradient: MK <= 31
                             directly, above as tap blocks of <= 31 x 31); single stages (ics_rl_stage) always the latter; env
                             ICS_CONV_PATH=vector|matrix|fft overrides the choice of AUTO                     */
#define ICS_CONV_VECTOR 1 /* packed-fp32 VALU convolutions (ics_conv.hip; ics_big.hip above 63) + fp32 PSF gradient: fp32 products */
#define ICS_CONV_MATRIX 2 /* fp16 MFMA kernels (ics_conv_mfma.hip MK <= 49, ics_gradk_mfma.hip MK <= 31): operands split
                             into two fp16 terms (22 significand bits), three MFMAs per product, fp32 accumulate */
#define ICS_CONV_FFT 3    /* transform tiles (ics_conv_fft.hip, round 5; since round 6 every MK: one tile to 85, tap blocks to 255): A1 / A3 / A11 as 128 x 128 overlap-save
                             FFTs held in LDS, fp32 throughout -- the reference's own method (scipy's complex64 FFT over the frame,
                             lib/deconvolution.pyx:478,491), tile by tile; against float64 direct sums 2 - 5e-7 of the largest
                             convolution value.  The frames live as channel-planar mirrors for the duration of a run; the PSF gradient
                             runs on the same tiles (two forward transforms per tile pair, products added up in the frequency domain; env
                             ICS_FFT_GRADK=0: on the matrix cores).  Round 6: A11 + A13 as one three-transform unit per tile pair
                             (ICS_FFT_FUSED=0: two kernels) and, up to 25 x 25, A1 + A3 as one unit whose interior tiles never leave the
                             frequency domain (ICS_FFT_CONV2=0: two kernels).  What ICS_CONV_AUTO picks inside ics_rl_run on large frames
                             (above); env ICS_CONV_PATH=fft forces it wherever it is built.
                             The PAM kinds (tv_mode 2, 3) run on it as well (TV term, back-projection epilogue G = T + lambd * gradu and
                             update on the planar mirrors); tv_mode 1 is refused.  Where image and u are exactly 0 the
                             transforms return rounding noise instead of exact zeros, like the reference's (see "DoF ratio" below) */

/* Accuracy of the matrix-core path (tests/test_gpu_precision.py drives it with adversarial inputs).  Every fp32 operand x of a
 * 78 x 80-pixel tile is scaled by a power of two s (tile maximum m -> [2^14, 2^15)) and split, x s = hi + lo + r, hi and lo
 * fp16.  fp16 carries 11 significand bits down to 2^-14 and a fixed quantum of 2^-24 below, so
 *        |r| / s  <=  max( 2^-22 |x| , 2^-39 m ):
 * 22 bits for every element within 2^17 of the tile maximum, an ABSOLUTE error of 2^-39 of the tile maximum for smaller
 * ones (the PSF taps likewise, against the largest tap).  Products drop the lo*lo term (< 2^-22 |x w|) and accumulate in fp32:
 *        |err(sum w x)|  <=  8 * 2^-22 * sum |w||x|  +  2^-38 * ( m * sum |w|  +  w_max * sum |x| ).
 * Measured: local relative error 0.6 - 1.4e-6 (the fp32 kernels: 0.7 - 2.5e-6) on ordinary frames, on 0..65535 frames and with
 * pixels 10^4 above their tile; for residuals of 1e-7 next to an isolated 1.0 the absolute error is 5e-13 (1.6e-5 of the
 * local values).  For comparison the reference's own convolution, scipy's FFT in complex64 (lib/deconvolution.pyx:478),
 * has an absolute error of ~1e-7 of the FRAME maximum at every pixel.  There is therefore no data-dependent fall-back to
 * the fp32 kernels in ICS_CONV_AUTO; ICS_CONV_VECTOR remains for callers who want fp32 products regardless. */

/* DoF ratio (lib/deconvolution.pyx:499, A5).  The mask is D = ((g - f)/(g + f))^2 with g the raw back-projection and f the image,
 * evaluated in IEEE float32 like the reference -- with ONE defined exception: where g == 0 and f == 0 EXACTLY the ratio is 1, not
 * 0/0 = NaN.  Reason: in a region where image and u are exactly black (clipped shadows, letterbox bars, zero borders of at least
 * 2 MK - 1 px) the reference's g is the rounding noise of scipy's complex64 FFT (~1e-10, either sign), and (g - 0)/(g + 0) is 1 for
 * EVERY non-zero g: the reference returns a finite picture there (22 of 24 black-row cases run with the compiled reference,
 * tests/golden/rl_black.npz) unless one noise value happens to be exactly 0, in which case its whole frame becomes NaN (the 2 other
 * cases, and most frames with black COLUMNS, whose noise is coarsely quantised).  This library's convolutions are exact sums:
 * g = 0 there, and IEEE 0/0 would lose every such frame through the next convolution and the NaN-propagating maxima.  With the
 * rule the results match the reference wherever the reference is finite (u <= 3e-7, psf <= 2e-7; tests/test_gpu_black.py), and
 * stay finite where it is not.  Nothing else is touched: g + f == 0 with g != 0 is +-inf as in the reference, and a NaN that is
 * already in the image or in u propagates (and is reported through ics_rl_stats.has_nan) exactly as before. */

#define ICS_TV_SHIPPED 0 /* lib/deconvolution.pyx as shipped: else-branches :519/:545        */
#define ICS_TV_MM_ACTIVE 1 /* BUILD-DEFINED extension, parity unpinned: the if-branches :517/:543 made
                              reachable (TV_ut from the majoriser, image denoising step :547-549 live);
                              exact definition in oracle/rl_ext_oracle.py.  `image` is modified.      */
#define ICS_TV_PAM_ISO 2   /* BUILD-DEFINED, parity unpinned: PAM u-step (Perrone & Favaro 2014; reference
                              README.md:42,106 prose only) with an isotropic TV gradient, PSF step as
                              pyx:555-589; no majoriser term, no DoF blend (oracle/rl_ext_oracle.py)  */
#define ICS_TV_PAM_COLLAB 3 /* same with the collaborative L-inf,1,1 RGB TV gradient (README.md:113-114) */

/* Scalars the reference only prints (pyx:593,648,659,665-669).  The per-outer-iteration traces go into CALLER-OWNED arrays
 * of `trace_cap` floats each (any of them may be NULL; trace_cap = params.iterations holds every iteration -- the reference
 * has no limit on the number of outer iterations and neither has this).  struct_size as in ics_rl_params; the library
 * preserves struct_size, trace_cap and the five pointers and overwrites everything else. */
typedef struct ics_rl_stats {
  uint32_t struct_size; /* in: sizeof(ics_rl_stats)                                        */
  int iterations_done; /* `it` at exit                                                    */
  int stopped;         /* stop_flag (1: the stop test fired, pyx:643-654); 2: the progress callback asked to stop */
  int has_nan;         /* np.any(np.isnan(u)) (pyx:671)                                   */
  float M_r, Hu, varu; /* values at exit (pyx:669)                                        */
  float dof_min, dof_max;
  int trace_len;       /* out: entries written = min(iterations_done, trace_cap)          */
  int trace_cap;       /* in                                                              */
  float *trace_M_r, *trace_Hu, *trace_varu, *trace_dof_min, *trace_dof_max; /* in: arrays of trace_cap floats or NULL */
  /* timing of the last ics_rl_run, measured with HIP events on the job's stream */
  float ms_total;      /* whole run (first launch -> last kernel), device time             */
  int inner_iterations;/* inner iterations executed                                       */
  /* params.profile = 1: average milliseconds per launch and launch count per kernel class */
  float ms_kernel[12];  /* ICS_KERNEL_COUNT */
  int launches[12];
} ics_rl_stats;
size_t ics_rl_stats_size(void);

/* Allocates the device frames for an M x N x 3 image and MK x MK x 3 PSF (MK odd, 3 <= MK <= 255):
 * u is (M+2*(MK/2)) x (N+2*(MK/2)) x 3 as in pyx:372-376.  All sizes run on the matrix cores (to 49 directly, above as tap
 * blocks; 65 ... 255: shipped loop only, tv_mode 0, fuse 0); fp32-product kernels behind ICS_CONV_VECTOR (to 127).  Limits (the reference has none; each
 * fails with ICS_ENOSUP and a message, never silently): PSF sizes above 255 (the reference's own examples go to 45,
 * deconvolve.py:409), stats windows
 * wider or higher than 4096 px (ics_rl_run; 8192-point transforms in 128 KB of LDS), frames of 2 GiB and more (32-bit
 * buffer offsets: about 13000 x 13000 px). */
int ics_rl_create(ics_ctx *ctx, int M, int N, int MK, ics_rl **out);
void ics_rl_destroy(ics_rl *job);
/* Host -> device.  image: M*N*3, u: uM*uN*3, psf: MK*MK*3 floats, C-contiguous HWC. */
int ics_rl_upload(ics_rl *job, const float *image, const float *u, const float *psf);
/* Device -> host; any pointer may be NULL.  `psf_caller` is what the reference leaves in the
 * CALLER's psf array (differs from the local psf when correlation != 0, pyx:585). */
int ics_rl_download(ics_rl *job, float *u, float *psf_local, float *psf_caller);
/* Runs the whole loop (pyx:460-659) on the device.  Host synchronisation happens once per outer
 * iteration (to read the stop-test scalars), never inside the 5 inner iterations. */
int ics_rl_run(ics_rl *job, const ics_rl_params *params, ics_rl_stats *stats);

/* Which kernels ics_rl_run will launch for this job and these parameters.  (ics_rl_stage takes the same route EXCEPT for the transform
 * tiles: single stages run on them only under an explicit params.conv = ICS_CONV_FFT with tv_mode 0 -- under ICS_CONV_AUTO, and for the PAM
 * kinds, a stage-driven loop such as lib/banded.py runs the matrix-core / vector kernels where ics_rl_run would take the tiles; describe
 * such a loop with params.conv = ICS_CONV_MATRIX.)  Inputs: (PSF size, params.conv, tv_mode, fuse,
 * flags, the ICS_CONV_PATH override): the library's own routing predicates, so that a benchmark labels its precision and traffic
 * figures from what actually runs.  Families:  convolutions A1/A3 -- 1 fp16-split matrix cores (whole PSF), 2 the same as tap blocks
 * (PSF > 49), 3 packed-fp32 kernels compiled per size, 4 run-time-sized fp32 kernels (ics_big.hip), 5 fp32 transform tiles on planar
 * mirrors (ics_conv_fft.hip), 6 the cooperative small-frame iteration kernel (ics_small.hip: the five inner iterations of an outer one in
 * ONE launch, a tile of a channel per compute unit, fp32 FMAs; frames up to ~290 px a side, PSF <= 31, shipped loop, ICS_CONV_AUTO only;
 * the PSF gradient is then family 8, inside the same launch);  PSF gradient A13 -- 1 fused with
 * A11 (k_synth_gradk, MK <= 15), 2 fp16-split matrix cores (k_gradk_mfma), 3 the same as tap blocks (MK >= 33), 4 fp32 MFMA
 * (k_gradk), 5 run-time-sized fp32 (k_gradk_big), 6 fp32 transform tiles (k_gradk_fft, ics_conv_fft.hip),
 * 7 fused with A11 on the fp32 transform tiles (k_synth_gradk_fft: three transforms per tile pair); 0 = not run (non-blind).  products_fp16_split = 1 when the products of that
 * stage are formed from two fp16 terms per operand (22 significand bits, fp32 accumulation), 0 = fp32 products. */
typedef struct ics_rl_route {
  uint32_t struct_size; /* in: sizeof(ics_rl_route) */
  int conv_family, conv_fp16_split;
  int gradk_family, gradk_fp16_split;
  int image_in_accumulator_order; /* the residual kernels read the read-only accumulator-order copy of the image (ics_image_acc.h) */
  int graph;                      /* ics_rl_run submits one hipGraph per outer iteration (small frames; ICS_GRAPH=0|1 overrides) */
} ics_rl_route;
int ics_rl_describe(ics_rl *job, const ics_rl_params *params, ics_rl_route *route);
/* The same for a shape alone -- no device and no job needed (a benchmark or a test labels its lines before anything is allocated). */
int ics_describe(int M, int N, int MK, const ics_rl_params *params, ics_rl_route *route);
/* Bytes ONE frame buffer of such a job takes on the device (aprons and tile padding included), 0 for an invalid shape: ics_rl_create refuses
   2 GiB and more (32-bit buffer offsets in the kernels).  No device needed.  lib/deconvolution.py uses it to send larger images through the row
   bands on one GPU (lib/banded.py) instead of failing -- the reference, lib/deconvolution.pyx:341, has no such limit. */
#define ICS_FRAME_LIMIT_BYTES 0x80000000ull
unsigned long long ics_rl_frame_bytes(int M, int N, int MK);

/* Stage-level entry points (parity tests, profiling).  They operate on the job's device frames. */
#define ICS_STAGE_SYNTH_RESIDUAL 1 /* A1+A2: error = conv_valid(u, psf) - image   (pyx:477-488)   */
#define ICS_STAGE_BACKPROJECT 2    /* A3 (+A7 reductions): gradu = corr_full(error, psf) (:490-491) */
#define ICS_STAGE_UPDATE 3         /* A5,A6,A8,A10: u update + DoF blend          (pyx:499-552)   */
#define ICS_STAGE_PSF_GRADIENT 4   /* A13: gradk                                   (pyx:567-571)   */
#define ICS_STAGE_PSF_UPDATE 5     /* A14-A17: psf step, tie, normalise, rotate    (pyx:574-589)   */
#define ICS_STAGE_MAJORIZE 6       /* ut = u                                       (pyx:462)       */
#define ICS_STAGE_STATS 7          /* A18+A19 on device -> stats scalars           (pyx:593-638)   */
#define ICS_STAGE_UPDATE_SYNTH 8   /* ICS_STAGE_UPDATE fused with ICS_STAGE_SYNTH_RESIDUAL (one kernel) */
#define ICS_STAGE_TVTERM 9         /* tv_mode 1: TV term T of u against ut (+ max|T_k|, max image_k)     */
#define ICS_STAGE_SYNTH_GRADK 10   /* A11 + A13 in one kernel (MK <= 15): gradk, and the residual e' over the whole frame */
/* One image over several jobs (row bands, lib/banded.py): a band job holds its rows plus a halo and computes everything on
 * them; only its OWNED rows are exact.  These two stages restrict the two global quantities of an inner iteration to them: */
#define ICS_STAGE_BAND_REDUCE 11   /* the step-size maxima of pyx:523-524 (max|g_k|, max u_k) over u-frame rows
                                      [band_row0, band_row1) only, from the back-projection of ICS_STAGE_BACKPROJECT: replaces
                                      the keys of ICS_BUF_RED; the caller combines them over the bands and writes them back */
#define ICS_STAGE_BAND_MASK_E 12   /* residual rows outside image rows [band_row0, band_row1) := 0, so that the PSF gradient
                                      of ICS_STAGE_PSF_GRADIENT sums the owned rows only (the caller adds the bands)       */
#define ICS_STAGE_SYNTH_BACKPROJECT 13 /* params.conv = ICS_CONV_FFT only: A1 + A2 + A3 (+A7) as ONE unit per tile pair (k_conv_fft<2>): gradu and the
                                          step-size maxima straight from u and the image; the residual frame is not written   */
int ics_rl_stage(ics_rl *job, int stage, const ics_rl_params *params);

/* Reads one device frame back in the reference's shape. */
#define ICS_BUF_U 0      /* uM x uN x 3  */
#define ICS_BUF_UT 1     /* uM x uN x 3  */
#define ICS_BUF_GRADU 2  /* uM x uN x 3 : raw back-projection (A3), before A6.  After ics_rl_run: the last inner iteration's on the multi-launch families;
                            on the transform tiles (family 5) it lives in the frame's planar mirror and the buffer read here is not refreshed; the
                            cooperative small-frame kernel (family 6) never stores it (it stays in LDS) -- single stages (ics_rl_stage) always write it */
#define ICS_BUF_IMAGE 3  /* M x N x 3    */
#define ICS_BUF_ERROR 4  /* M x N x 3    */
#define ICS_BUF_PSF 5    /* MK x MK x 3  */
#define ICS_BUF_GRADK 6  /* MK x MK x 3  */
#define ICS_BUF_TV 8      /* uM x uN x 3 : TV term T (tv_mode 1) */
#define ICS_BUF_SCALARS 7 /* 16 floats: dt[3], maxu[3], maxg[3], dtpsf, M_r, Hu, varu, dof_min, dof_max, 0 */
#define ICS_BUF_RED 9     /* 16 words (bit patterns in float slots): reduction keys of stage calls -- [0..2] max|g_k|,
                             [3..5] max u_k as order-preserving uint32 keys (larger key = larger float; NaN = 0xFFC00000);
                             read only: [12] min key, [13] max key, [14] NaN flag of the DoF mask of the last update (pyx:593) */
int ics_rl_read(ics_rl *job, int which, float *host, size_t count);
int ics_rl_write(ics_rl *job, int which, const float *host, size_t count);
/* Rows [row0, row0 + nrows) of a frame buffer (ICS_BUF_U / UT / GRADU: uN*3 floats per row; IMAGE / ERROR: N*3). */
int ics_rl_read_rows(ics_rl *job, int which, int row0, int nrows, float *host);
int ics_rl_write_rows(ics_rl *job, int which, int row0, int nrows, const float *host);
/* Rows of a frame buffer of one job into a frame buffer of another, device to device (the halo exchange and the stop-test
   gather of the row-band split, lib/banded.py): the two frames must have the same row length.  Jobs on different GPUs need peer
   access (xGMI); ICS_ENOSUP if the devices cannot reach each other -- the caller then goes through the host. */
int ics_rl_copy_rows(ics_rl *dst, int dst_which, int dst_row0, ics_rl *src, int src_which, int src_row0, int nrows);

/* Kernel classes indexing ics_rl_stats.ms_kernel / launches. */
#define ICS_K_SYNTH 0        /* A1+A2 convolution kernel        */
#define ICS_K_BACKPROJECT 1  /* A3 correlation kernel (+A7)     */
#define ICS_K_UPDATE 2       /* A5/A6/A8/A10 elementwise kernel */
#define ICS_K_PSF_GRADIENT 3 /* A13 MFMA kernel (+ reduction)   */
#define ICS_K_PSF_UPDATE 4   /* A14-A17                         */
#define ICS_K_MAJORIZE 5     /* ut = u copy                     */
#define ICS_K_STATS 6        /* A18/A19 window statistics + FFT */
#define ICS_K_UPDATE_SYNTH 7 /* fused A5-A10 + A1/A2 (or A11) kernel */
#define ICS_K_SYNTH_GRADK 8  /* fused A11 + A13 kernel (+ reduction)  */
#define ICS_K_SYNTH_BACKPROJECT 9 /* A1 + A2 + A3 (+A7) in one unit per tile pair (transform tiles, small PSFs) */
#define ICS_K_SMALL_ITER 10   /* small frames: the inner iterations of an outer one as ONE cooperative launch (ics_small.hip) */
#define ICS_KERNEL_COUNT 12  /* (11 reserved) */

/* ---- small standalone operators ---------------------------------------------------------- */
/* lib/deconvolution.pyx:73-75 -- in place on a host MK*MK*3 float32 array, computed on device. */
int ics_normalize_kernel(ics_ctx *ctx, float *kern, int MK);
/* lib/deconvolution.pyx:137-239 -- u: M*N*3 -> out, div (borders untouched = 0). */
int ics_tv(ics_ctx *ctx, const float *u, int M, int N, float epsilon, int order, int norm, float *out, float *div);
/* lib/utils.py:237-264 -- scipy.signal.convolve2d(src, kern, mode="same", boundary="symm"), float64.  LDS-tiled; a rank-1
 * kernel (every lib/utils.py window is one) runs as a row pass and a column pass.  The three filters keep their device
 * scratch in the context between calls. */
int ics_conv2d_symm(ics_ctx *ctx, const double *src, int H, int W, const double *kern, int KH, int KW, double *out);
/* lib/utils.py:267-277 -- src + (src - conv2d_symm(src, kern)) * amount. */
int ics_usm(ics_ctx *ctx, const double *src, int H, int W, const double *kern, int KH, int KW, double amount, double *out);
/* lib/utils.py:173-234 -- bilateral filter, symmetric padding, gaussian(x, s) = exp(-x^2/(2 s^2)). */
int ics_bilateral(ics_ctx *ctx, const double *src, int H, int W, int radius, double std_i, double std_s, double *out);

/* deconvolve.py:245-249 -- src: H x W x C float64 (HWC) -> out: OH x OW x C.  Gaussian anti-aliasing with
 * sigma = (scale - 1) / 2 per axis when shrinking, cubic B-spline interpolation at the pixel-centre grid, edge mode
 * "nearest"; the algorithm is written out in oracle/resize_oracle.py and pinned there against scipy.ndimage. */
int ics_resize_bicubic(ics_ctx *ctx, const double *src, int H, int W, int C, double *out, int OH, int OW);

/* ---- device-resident images -----------------------------------------------
 * H x W x 3 float32 HWC images in HBM, so that deblur_module (deconvolve.py:65-368) keeps its frames on the
 * device across pyramid levels and between the blind and the non-blind phase.  Operations are queued on the
 * context's stream; upload / download synchronise.  Functions with an `out` argument create a new image. */
typedef struct ics_img ics_img;
int ics_img_create(ics_ctx *ctx, int H, int W, ics_img **out);
void ics_img_destroy(ics_img *img);
int ics_img_shape(const ics_img *img, int *H, int *W);
int ics_img_upload(ics_img *img, const float *host);        /* H*W*3 floats */
/* the same from 8- or 16-bit pixels (what deconvolve.py:357-368 reads from its files): bytes_per_value = 1 (uint8) or 2 (uint16); the
 * values cross PCIe as they are and become float32 on the device (exact: np.float32(v)) -- a quarter / half of the float transfer and
 * no host-side conversion pass */
int ics_img_upload_int(ics_img *img, const void *host, int bytes_per_value);
int ics_img_download(const ics_img *img, float *host);
/* deconvolve.py:24-37 pad_image (np.pad mode="edge" on the two spatial axes) */
int ics_img_pad_edge(const ics_img *src, int top, int bottom, int left, int right, ics_img **out);
/* src[y0:y0+H, x0:x0+W] (the slices of deconvolve.py:322-323, :360-366) */
int ics_img_crop(const ics_img *src, int y0, int x0, int H, int W, ics_img **out);
/* dst[y0:y0+h, x0:x0+w] = src */
int ics_img_paste(ics_img *dst, int y0, int x0, const ics_img *src);
/* in place: x = powf(clip01 ? clip(x / div, 0, 1) : x / div, exponent) * mul
 * (deconvolve.py:100-103: div = 2^bits - 1, exponent = 1/2.2; :346-352: clip, exponent = 2.2, mul = 65535) */
int ics_img_gamma(ics_img *img, float div, float exponent, float mul, int clip01);
/* deconvolve.py:245-249 on a device image (ics_resize_bicubic semantics, result rounded to float32) */
int ics_img_resize(const ics_img *src, int OH, int OW, ics_img **out);
/* richardson_lucy_MM(image[iy:iy+M, ix:ix+N], u[uy:uy+uM, ux:ux+uN], psf, ...) with both windows taken from device
 * images (deconvolve.py:277-313 passes such views); psf is a host MK*MK*3 array as in ics_rl_upload. */
int ics_rl_upload_img(ics_rl *job, const ics_img *image, int iy, int ix, const ics_img *u, int uy, int ux, const float *psf);
/* The reference updates the caller's `u` view in place, border ring included (pyx:527-531): the whole u frame goes
 * back to dst[y:y+uM, x:x+uN]. */
int ics_rl_download_img(ics_rl *job, ics_img *dst, int y, int x);

/* ---- one image per GPU (BASELINE.json configs[4]) ------------------------------------------------
 * The reference has no multi-device code: every richardson_lucy_MM call (lib/deconvolution.pyx:341) is an independent
 * job, so N GPUs run N jobs, one process per GPU, and nothing is exchanged during the iterations.  The only collective
 * is the gather of a small per-rank record at the end -- RCCL over xGMI, called directly from this library (librccl.so
 * is dlopen'ed by ics_group_create when world > 1).  `rendezvous` is a file path shared by the ranks of the node: rank
 * 0 publishes the RCCL unique id there, the others wait up to `timeout_s` seconds for it.  With world == 1 every call
 * is a local no-op / copy and no device is touched.  count <= 49152 doubles per call (3 x 127^2 fits). */
typedef struct ics_group ics_group;
int ics_group_create(int device, int rank, int world, const char *rendezvous, int timeout_s, ics_group **out);
void ics_group_destroy(ics_group *g);
int ics_group_info(const ics_group *g, int *rank, int *world);
int ics_group_barrier(ics_group *g);
int ics_group_allreduce_max(ics_group *g, double *inout, int count);
int ics_group_allreduce_sum(ics_group *g, double *inout, int count); /* summed in rank order by RCCL (ncclSum, float64) */
/* How the group is connected: *backend = 0 local (world 1, no communicator), 1 RCCL; *nranks = ranks RCCL reports for the
 * communicator (ncclCommCount), lib = path or soname of the RCCL library in use ("" when local). */
int ics_group_describe(const ics_group *g, int *backend, int *nranks, char *lib, size_t lib_len);
int ics_group_allgather(ics_group *g, const double *send, int count, double *recv /* world * count */);
/* One image over several ranks (row bands, one process per GPU: lib/banded.py rank mode).  Rows [send_row0, + send_rows) of frame
 * buffer `which` of this rank's band job go to rank send_peer, rows [recv_row0, + recv_rows) arrive from rank recv_peer (a peer
 * < 0 switches that side off): RCCL point-to-point over xGMI between the two jobs' device frames, both directions in one group
 * call so that neighbours exchanging halos cannot deadlock.  Synchronous: returns when the rows are in place. */
int ics_rl_exchange_rows(ics_rl *job, ics_group *g, int which, int send_row0, int send_rows, int send_peer, int recv_row0, int recv_rows, int recv_peer);
/* The two per-iteration reductions of the row-band split, IN PLACE on the band job's device buffers and on the job's own stream -- one
 * RCCL call each, no host staging, no synchronisation (round 3 moved them through the host in chunks of 64 doubles):
 *   ics_rl_allreduce_keys   the six step-size keys of ICS_BUF_RED [0..5] (order-preserving uint32 keys: ncclMax on ncclUint32 is exact);
 *   ics_rl_allreduce_gradk  the 3 MK^2 PSF-gradient sums of ICS_BUF_GRADK: widened to float64 on the device, summed over the ranks
 *                           (ncclSum, float64), rounded to float32 once.
 * With a one-rank local group both return at once. */
int ics_rl_allreduce_keys(ics_rl *job, ics_group *g);
int ics_rl_allreduce_gradk(ics_rl *job, ics_group *g);

#ifdef __cplusplus
}
#endif
#endif /* ICS_HIP_H */
