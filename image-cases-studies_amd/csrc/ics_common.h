// ics_common.h -- shared definitions for the gfx950 kernels of libics_hip.so.
//
// Device data layout ("frames").  Every full-size array of the reference loop
// (lib/deconvolution.pyx:378-390: u, ut, gradu, image, error) lives in one common geometry,
// expressed in the coordinates of `u` (the "u-frame", (M+2pad) x (N+2pad) x 3, HWC fp32):
//
//        <- ax px -><------------ uN px (tiles of 64) ------------->< slack ><- ax ->
//   ay rows of zeros (ay >= pad)
//   +----------------+--------------------------------------------------------------+
//   | apron (zeros)  |  u-frame rows 0..uM-1; the image/error live at (+pad,+pad)   |
//   ...
//   slack rows up to a multiple of 64, then ay rows of zeros
//
//   * pitch (floats per row) is a multiple of 64 floats (256 B) and ax is a multiple of 4 px, so
//     every 64-px tile row starts 16-B aligned for dwordx4 traffic although a pixel is 12 B.
//   * the apron is >= pad on every side and is never written, so convolution tiles load their
//     halo without bounds checks and the `full` back-projection (pyx:491) sees its zero
//     extension for free.
//   * `image`, `error` use the same geometry shifted by (+pad,+pad) and are zero outside M x N.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>

#define ICS_TILE 64 /* tile edge in pixels; frames are padded to a multiple of it */

struct IcsGeom {
  int M, N;      // image size
  int K, pad;    // PSF size and K/2
  int uM, uN;    // u-frame size = M+2pad, N+2pad
  int ax, ay;    // apron: ax px left/right (multiple of 4, >= pad), ay rows top/bottom (>= pad)
  int pitch;     // floats per row
  int rows;      // allocated rows
  int tiles_x, tiles_y;
  int wrow;      // floats per packed weight row = round_up(6*K, 4); K+1 rows (see IcsConvArgs::w)
};

static inline IcsGeom ics_make_geom(int M, int N, int K) {
  IcsGeom g;
  g.M = M; g.N = N; g.K = K; g.pad = K / 2;
  g.uM = M + 2 * g.pad; g.uN = N + 2 * g.pad;
  // 16*nb: the PSF-gradient kernel (ics_kernels.hip, k_gradk) evaluates 16x16 blocks of taps with
  // MFMA and therefore touches up to 16*nb rows/pixels around a tile
  const int nb = (K + 15) / 16;
  g.ax = (g.pad + 3) & ~3; if (g.ax < 16 * nb) g.ax = 16 * nb;
  g.ay = 16 * nb;
  g.tiles_x = (g.uN + ICS_TILE - 1) / ICS_TILE;
  g.tiles_y = (g.uM + ICS_TILE - 1) / ICS_TILE;
  int px = g.ax + g.tiles_x * ICS_TILE + g.ax;
  g.pitch = ((3 * px + 63) / 64) * 64;
  g.rows = g.ay + g.tiles_y * ICS_TILE + g.ay + 1;  // +1: slack row for tiles that over-read
  g.wrow = (6 * K + 3) & ~3;
  return g;
}
static inline size_t ics_frame_floats(const IcsGeom& g) { return (size_t)g.rows * g.pitch; }
// Channel-planar mirror of a frame (operands of the FFT convolutions, ics_conv_fft.hip): three planes of g.rows rows, one float per pixel,
// same aprons and tile slack as the HWC frame; plane row = the HWC row's pixel count rounded up to 64 floats (never more than a third of the
// HWC pitch rounded up: a planar mirror fits the HWC frame's allocation + 3 * 64 * rows floats).  Pixel (y, x, c) of the u-frame sits at
// plane_origin + c * plane_floats + y * ppitch + x.
__host__ __device__ static inline int ics_ppitch(const IcsGeom& g) { return (g.ax + g.tiles_x * ICS_TILE + g.ax + 63) / 64 * 64; }
__host__ __device__ static inline size_t ics_plane_floats(const IcsGeom& g) { return (size_t)g.rows * ics_ppitch(g); }
__host__ __device__ static inline size_t ics_planar_floats(const IcsGeom& g) { return 3 * ics_plane_floats(g); }
__host__ __device__ static inline size_t ics_planar_origin(const IcsGeom& g) { return (size_t)g.ay * ics_ppitch(g) + g.ax; }
static inline size_t ics_origin_offset(const IcsGeom& g) { return (size_t)g.ay * g.pitch + 3 * (size_t)g.ax; }

// ---- per-device launcher state ----------------------------------------------------------------------------------------
// The launchers keep two things per device: the compute-unit count and "the dynamic-LDS attribute of this kernel is set".
// lib/banded.py drives one job per band from one host thread each, so these are atomics; the attribute call is idempotent, a
// lost race only repeats it.
#define ICS_MAX_DEVICES 64
static inline int ics_current_device() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= ICS_MAX_DEVICES) dev = 0;
  return dev;
}
static inline int ics_device_cus(int dev) {
  static std::atomic<int> cus[ICS_MAX_DEVICES];   // static storage: zero
  int n = cus[dev].load(std::memory_order_relaxed);
  if (!n) {
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) { (void)hipGetLastError(); n = 256; }
    cus[dev].store(n, std::memory_order_relaxed);
  }
  return n;
}
// `done` = one std::atomic<bool> per device, static in the launcher (one array per kernel instance)
template <typename Kern>
static inline hipError_t ics_configure_lds(std::atomic<bool>* done, int dev, Kern kern, size_t bytes) {
  if (done[dev].load(std::memory_order_acquire)) return hipSuccess;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) { (void)hipGetLastError(); return e; }   // do not leave a sticky error behind
  done[dev].store(true, std::memory_order_release);
  return hipSuccess;
}

// ---- process-wide switches for tests and A/B measurements (ics_debug.h; NOT part of include/ics_hip.h) -------------------
// Read from the environment ONCE (first use), changed at run time through ics_debug_set(): no getenv on any launch path.
struct IcsDebug {
  std::atomic<int> max_wgs;           // ICS_TEST_MAX_WGS      cap on persistent workgroups, 0 = none (tests: many tiles per workgroup)
  std::atomic<int> dynamic_tiles;     // ICS_DYNAMIC_TILES     -1 launcher decides, 0 static walk, 1 dynamic tile claiming
  std::atomic<int> conv_rs;           // ICS_TEST_CONV_RS      0 launcher decides, 2 / 4 = force 32- / 64-row tiles where both are built
  std::atomic<int> conv_nh;           // ICS_TEST_CONV_NH      harness builds only: 1 = 4-wave form of the K >= 23 kernels
  std::atomic<int> conv_path;         // ICS_CONV_PATH         what ICS_CONV_AUTO resolves to: 0 default, 1 vector, 2 matrix, 3 fft (transform tiles, ics_conv_fft.hip)
  std::atomic<int> fused_gradk;       // ICS_FUSED_GRADK       0 = two-kernel A11 + A13 (like ICS_FLAG_NO_FUSED_GRADK)
  std::atomic<int> update_wg_per_cu;  // ICS_UPDATE_WG_PER_CU  0 launcher decides
  std::atomic<int> update_kernel;     // ICS_UPDATE_KERNEL     0 = the pixel-group kernel everywhere
  std::atomic<int> fused_rs;          // ICS_FUSED_RS          0 launcher decides, 2 / 4 = tile height of the fused A11 + A13 kernel
  std::atomic<int> planar_image;      // ICS_PLANAR_IMAGE      0 = epilogues read the HWC image frame (no accumulator-order copy)
  std::atomic<int> pam_exact;         // ICS_PAM_EXACT         1 = the TV term of ALL three extended kinds (tv_mode 1, 2, 3) with IEEE sqrt / division per value
  std::atomic<int> fail_window_alloc; // (test hook)           n = the n-th allocation of the next ensure_window() fails once with ICS_ENOMEM
  std::atomic<int> pool_limit_mb;     // ICS_POOL_LIMIT_MB     cap of a context's cache of freed device blocks in MiB (-1: a quarter of the device memory; 0: no caching)
  std::atomic<int> overlap;           // ICS_OVERLAP           0 = drain at every outer boundary, 1 (default) = statistics on a second stream where that measured ahead (use_overlap, ics_api.hip), 2 = second stream always, 3 = look-ahead with the statistics on the job's own stream
  std::atomic<int> fft_gradk;         // ICS_FFT_GRADK         0 = the FFT-tile pipeline takes its PSF gradient on the matrix cores (k_gradk_mfma on the mirrors) instead of on the tiles
  std::atomic<int> fft_fused;         // ICS_FFT_FUSED         0 = the FFT-tile pipeline runs A11 and A13 as two kernels (k_conv_fft<0> + k_gradk_fft) instead of the fused three-transform unit
  std::atomic<int> fft_conv2;         // ICS_FFT_CONV2         0 = the FFT-tile pipeline runs A1 and A3 as two kernels; 1 (default) = as one unit per tile pair (k_conv_fft<2>) for the PSF sizes it pays for; 2 = wherever it is built
  std::atomic<int> fft_rot;           // ICS_FFT_ROT           0 = mode 2 of the tiles walks its units from the first tile row (the last, partial round is then the bottom row's four-transform units)
  std::atomic<int> small_iter;        // ICS_SMALL_ITER        0 = small frames run the multi-launch families instead of the cooperative iteration kernel (ics_small.hip); 2 = 64-pixel tiles too; default 1, 0 under rocprofv3
  std::atomic<int> fail_small_launch; // (test hook)           1 = the next cooperative launch of the small-frame kernel is refused once (the job falls back to the multi-launch path)
  std::atomic<int> small_trace;       // ICS_SMALL_TRACE       1 = every cooperative launch is followed by a drain and a phase timeline on stderr
  std::atomic<int> graph;             // ICS_GRAPH             0 (default) never, 1 always, -1 frames <= 1.2 Mpx: one hipGraph launch per outer iteration (measured: no gain, NOTES_r04.md 4d)
  static int env_int(const char* name, int dflt) { const char* e = getenv(name); return (e && e[0]) ? atoi(e) : dflt; }
  IcsDebug() {
    max_wgs = env_int("ICS_TEST_MAX_WGS", 0);
    dynamic_tiles = env_int("ICS_DYNAMIC_TILES", -1);
    conv_rs = env_int("ICS_TEST_CONV_RS", 0);
    conv_nh = env_int("ICS_TEST_CONV_NH", 0);
    const char* cp = getenv("ICS_CONV_PATH");
    conv_path = !cp ? 0 : (cp[0] == 'v' ? 1 : (cp[0] == 'm' ? 2 : (cp[0] == 'f' ? 3 : 0)));
    fused_gradk = env_int("ICS_FUSED_GRADK", 1);
    update_wg_per_cu = env_int("ICS_UPDATE_WG_PER_CU", 0);
    update_kernel = env_int("ICS_UPDATE_KERNEL", 1);
    fused_rs = env_int("ICS_FUSED_RS", 0);
    planar_image = env_int("ICS_PLANAR_IMAGE", 1);
    pam_exact = env_int("ICS_PAM_EXACT", 0);
    fail_window_alloc = 0;
    fft_gradk = env_int("ICS_FFT_GRADK", 1);
    fft_fused = env_int("ICS_FFT_FUSED", 1);
    fft_conv2 = env_int("ICS_FFT_CONV2", 1);
    fft_rot = env_int("ICS_FFT_ROT", 1);
    // (a process that has made a cooperative launch under rocprofv3 -- ROCm 7.2, rocprofiler-sdk tool library preloaded -- crashes in its exit handlers AFTER the
    //  profiler has written its output: exit code 139 for an otherwise complete run.  Under an attached profiler the small frames therefore stay on the
    //  multi-launch families unless ICS_SMALL_ITER=1 asks for the cooperative kernel explicitly)
    const char* tool = getenv("ROCP_TOOL_LIBRARIES");
    const char* pre = getenv("LD_PRELOAD");
    const bool profiler = (tool && tool[0]) || (pre && strstr(pre, "rocprofiler"));
    small_iter = env_int("ICS_SMALL_ITER", profiler ? 0 : 1);
    small_trace = env_int("ICS_SMALL_TRACE", 0);
    fail_small_launch = 0;
    graph = env_int("ICS_GRAPH", 0);
    overlap = env_int("ICS_OVERLAP", 1);
    pool_limit_mb = env_int("ICS_POOL_LIMIT_MB", -1);
  }
};
// one instance per process (inline function, function-local static: initialised once, thread-safe)
inline IcsDebug& ics_debug() { static IcsDebug d; return d; }
// extern "C" int ics_debug_set(const char* name, int value) -- exported by libics_hip.so for the tests (lib/_native.py
// debug_set), declared here only: names = the lower-case field names above; returns 0, or -1 for an unknown name

// Order-preserving float <-> uint32 key so that atomicMax/atomicMin on the key is a float max/min.
// Key 0 is below every float (used as the "empty" value for max), 0xFFFFFFFF above (for min).
__host__ __device__ static inline uint32_t ics_f2key(float f) {
  union { float f; uint32_t u; } v; v.f = f;
  return (v.u & 0x80000000u) ? ~v.u : (v.u | 0x80000000u);
}
__host__ __device__ static inline float ics_key2f(uint32_t k) {
  union { float f; uint32_t u; } v;
  v.u = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
  return v.f;
}

// The ratio of the DoF mask, (g - f)/(g + f)  (lib/deconvolution.pyx:499; g = raw back-projection, f = image), IEEE in every
// case but one: g == f == 0 EXACTLY gives 1, not 0/0 = NaN.  Where image and u are exactly black the reference's g is the rounding
// noise of its complex64 FFT (~1e-10, either sign) and (g - 0)/(g + 0) == 1 for every non-zero g, so the reference returns a
// finite picture on frames with black bands; this library's convolutions return exact zeros there, and a single NaN would spread
// over the whole frame through the next convolution and the NaN-propagating maxima.  g + f == 0 with g != 0 stays +-inf as in
// the reference.  Contract and measurements: include/ics_hip.h ("DoF ratio"), tests/test_gpu_black.py.
__device__ __forceinline__ float ics_dof_ratio(float g, float f) {
  const float d = __fdiv_rn(__fsub_rn(g, f), __fadd_rn(g, f));
  return (g == 0.f && f == 0.f) ? 1.0f : d;
}

// Persistent tile walks (ics_conv_mfma.hip, ics_synth_gradk_mfma.hip): workgroup b runs on XCD b % nb and walks the band of tiles
// [ics_band_begin(x), ics_band_begin(x + 1)) of its XCD x.  A band's share of the tiles follows the number of workgroups that walk it
// (grid / nb, or one more for the first grid % nb XCDs).  With equal shares, 85 tiles dealt to 85 workgroups gave two XCDs 11 tiles
// and 10 workgroups: one workgroup walked two tiles and a 255^2 back-projection took 16 us instead of 10 (phase timeline,
// tools/bench_conv_mfma.hip -DICS_MFMA_TRACE).
__host__ __device__ inline int ics_band_begin(int ntiles, int grid, int nb, int x) {
  const int q = grid / nb, r = grid - q * nb;
  return (int)((long)ntiles * (q * x + (x < r ? x : r)) / grid);
}

// Fair shares for the workgroups that share a CU under a STATIC tile walk (ics_synth_gradk_mfma.hip, ics_gradk_mfma.hip; the walks are
// static because a workgroup's partial sums must not depend on timing).  The CU's arbiter serves the oldest wave first: of the two
// workgroups of a CU (blocks b and b + CUs of a grid of 2 x CUs) the first-dispatched one walked its tiles a quarter faster and left its
// mate alone on a half-empty CU for the last fifth of the kernel.  Priority alternates between the mates in slices of 2^slice ticks of the
// 100 MHz wall clock; `team` = which mate (blockIdx / (grid / mates)).  Scheduling only -- results do not change.
__device__ __forceinline__ void ics_prio_turn(int slice, int team, int nteams) {
  if (slice <= 0) return;
  const unsigned turn = (unsigned)(wall_clock64() >> slice) % (unsigned)nteams;
  if (turn == (unsigned)team) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
}

// Wave-wide maximum (all 64 lanes receive it) without ds_bpermute: four DPP steps inside each row of 16 lanes, then the four
// row results through v_readlane.  The shuffle form (__shfl_xor = ds_bpermute) needs one address register per step; inside a
// persistent tile loop the compiler hoisted those addresses above the loop and, in the 256-register kernels, spilled one of
// them -- a scratch reload that waits on vmcnt, i.e. on the whole next-tile prefetch in flight.  maxnum semantics (NaN dropped).
__device__ __forceinline__ float ics_wave_max_f32(float v) {
#define ICS_DPP_F32(x, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (x)), (ctrl), 0xF, 0xF, true))
  v = __builtin_fmaxf(v, ICS_DPP_F32(v, 0xB1));    // quad_perm [1, 0, 3, 2]
  v = __builtin_fmaxf(v, ICS_DPP_F32(v, 0x4E));    // quad_perm [2, 3, 0, 1]
  v = __builtin_fmaxf(v, ICS_DPP_F32(v, 0x141));   // row_half_mirror
  v = __builtin_fmaxf(v, ICS_DPP_F32(v, 0x140));   // row_mirror
#undef ICS_DPP_F32
  const int iv = __builtin_bit_cast(int, v);
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 48));
  return __builtin_fmaxf(__builtin_fmaxf(r0, r1), __builtin_fmaxf(r2, r3));
}
__device__ __forceinline__ uint32_t ics_wave_max_u32(uint32_t v) {
#define ICS_DPP_U32(x, ctrl) (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(x), (ctrl), 0xF, 0xF, true)
  uint32_t o;
  o = ICS_DPP_U32(v, 0xB1); v = v > o ? v : o;
  o = ICS_DPP_U32(v, 0x4E); v = v > o ? v : o;
  o = ICS_DPP_U32(v, 0x141); v = v > o ? v : o;
  o = ICS_DPP_U32(v, 0x140); v = v > o ? v : o;
#undef ICS_DPP_U32
  const uint32_t r0 = (uint32_t)__builtin_amdgcn_readlane((int)v, 0), r1 = (uint32_t)__builtin_amdgcn_readlane((int)v, 16);
  const uint32_t r2 = (uint32_t)__builtin_amdgcn_readlane((int)v, 32), r3 = (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
  const uint32_t a = r0 > r1 ? r0 : r1, b = r2 > r3 ? r2 : r3;
  return a > b ? a : b;
}

// Reduction slots written by the back-projection kernel (one set per inner iteration).
#define ICS_RED_MAXG 0 /* 3 keys: max |gradu_k| after A6 (pyx:524)  */
#define ICS_RED_MAXU 3 /* 3 keys: max u_k                (pyx:524)  */
#define ICS_RED_MAXT 6 /* 3 keys: max |T_k|  (active MM-TV: gradu of pyx:543)          */
#define ICS_RED_MAXF 9 /* 3 keys: max image_k                (pyx:548)          */
#define ICS_RED_STRIDE 16

// Device scalar block (floats), mirrored by ICS_BUF_SCALARS in include/ics_hip.h.
#define ICS_SC_DT 0
#define ICS_SC_MAXU 3
#define ICS_SC_MAXG 6
#define ICS_SC_DTPSF 9
#define ICS_SC_MR 10
#define ICS_SC_HU 11
#define ICS_SC_VARU 12
#define ICS_SC_DOFMIN 13
#define ICS_SC_DOFMAX 14
#define ICS_SC_COUNT 16

// ---- launchers implemented in the .hip translation units ------------------------------------
struct IcsConvArgs {
  const float* in;   // frame origin of the convolved array (u for A1, error for A3)
  const float* w;    // packed weights, correlation orientation W[a][b][c], as ROW PAIRS for v_pk_fma_f32:
                     // w[ap][ (3b+c)*2 + h ] = W[ap - h][b][c]  (ap = 0..K, h = 0/1, W[-1] = W[K] = 0)
  float* out;        // frame origin of the output (error for A1, gradu for A3)
  const float* f;    // A1: image frame origin
  const float* u;    // A3: u frame origin   (for the fused A6/A7 reductions)
  const float* ut;   // A3: ut frame origin
  const float* tv;   // A3, extended modes only: T frame (else NULL), see k_tvterm
  int tv_kind;       // 0 shipped, 1 active MM-TV, 2/3 PAM (isotropic / collaborative TV)
  uint32_t* red;     // A3: reduction keys (ICS_RED_*) written; mode 2: keys of the finished back-projection read
  float lambd;
  // mode 2 only (update of the previous inner iteration fused in front of the convolution):
  const float* gr;   // raw back-projection frame (A3 output)
  float* u_out;      // frame that receives the updated u (ping-pong partner of `in`)
  float* scal;       // device scalar block: dt / maxu / maxg recorded
  uint32_t* dofkeys; // DoF min/max/NaN keys (when want_dof)
  float step;
  int blind, want_dof;
  const void* bt;    // matrix-core path only: Toeplitz fragment table of this orientation (ics_conv_mfma.hip), else NULL
  const float* facc[2];  // matrix-core path, mode 0: the image in accumulator order for 32-row ([0], RS = 2) and 64-row ([1], RS = 4)
                         // tiles (ics_image_acc.h); NULL = the epilogue reads the HWC frame
  uint32_t* sched;   // matrix-core path: 9 zeroed words for the dynamic tile walk (8 per-band claim counters + 1 exit counter;
                     // the last workgroup to leave zeroes them again), or NULL = static interleaved walk
  IcsGeom g;
};
// mode 0 = A1+A2 (valid convolution + residual), mode 1 = A3 (+A6/A7 reductions),
// mode 2 = A5/A6/A8/A10 (update, recomputed on the halo) + A1+A2 on the updated u
hipError_t ics_launch_conv(int mode, const IcsConvArgs& a, hipStream_t s);
bool ics_conv_supported(int K);
// Matrix-core variant (ics_conv_mfma.hip): modes 0 and 1, odd K <= 37, operands split into two fp16 terms.
// Weight table (built by k_psf, = the kernel's LDS image): row c*K + a of 2*WROWB bytes, WROWB = round4(2*(K+17)), holds
// halves 8.. of the scaled zero-padded kernel row Wp[idx] = s_w * W[a][idx - 15][c] (taps at local halves 7 .. K+6) as its
// two fp16 split terms interleaved dword by dword (hi dword d at 2d, lo dword d at 2d + 1); one float 1/s_w behind the last row.
hipError_t ics_launch_conv_mfma(int mode, const IcsConvArgs& a, hipStream_t s);
bool ics_conv_mfma_supported(int K);
bool ics_conv_mfma_preferred(int K);   // what ICS_CONV_AUTO picks
size_t ics_conv_mfma_table_floats(int K);
int ics_conv_mfma_rs(int K, const IcsGeom& g, int cus = -1);   // 2 / 4: tile height mode 0 will run with (accumulator-order image layout), 0: none
