// ics_api.hip -- the C ABI of libics_hip.so (include/ics_hip.h): contexts, the device-resident
// Richardson-Lucy job, the outer/inner iteration schedule of lib/deconvolution.pyx:460-659, and the
// standalone operators.  Host side only; kernels live in ics_conv.hip / ics_kernels.hip /
// ics_stats.hip / ics_filters.hip.
//
// Schedule of one inner iteration (all on one HIP stream, no host round trip):
//   k_conv<mode 0>  A1+A2   error = conv(u, psf) - image
//   k_conv<mode 1>  A3+A7   gradu = corr(error, psf); max|g_k|, max u_k  -> device keys
//   k_update        A5-A10  dt on device from the keys; u update + DoF blend
//   blind only: k_conv<mode 0> (A11), k_gradk + k_gradk_reduce (A13, MFMA), k_psf (A14-A17)
// Once per outer iteration: ut = u (D2D), window statistics + FFT whiteness metric (A18/A19), one
// 64-byte D2H of the scalars, and the stop decision on the host (pyx:643-654).
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <new>
#include <vector>
#include <map>
#include <mutex>
#include <unordered_map>

#include "../../include/ics_hip.h"
#include "ics_kernels.h"
#include "ics_image_acc.h"

// -------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
static int fail(int code, const char* fmt, ...) {
  va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap);
  return code;
}
// the same for the other translation units (ics_group.hip)
int ics_set_error(int code, const char* fmt, ...) {
  va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap);
  return code;
}
#define HIPCHK(x)                                                                                \
  do {                                                                                           \
    hipError_t e_ = (x);                                                                         \
    if (e_ != hipSuccess)                                                                        \
      return fail(e_ == hipErrorOutOfMemory ? ICS_ENOMEM : ICS_EHIP, "%s failed: %s (%s:%d)", #x, \
                  hipGetErrorString(e_), __FILE__, __LINE__);                                    \
  } while (0)

#define RC(x) do { int rc_ = (x); if (rc_ != ICS_OK) return rc_; } while (0)

// Device memory of a context is recycled, not returned (round 4).  deblur_module creates a job and a handful of images per pyramid
// level and phase (deconvolve.py:204-313); hipMalloc / hipFree cost 0.1 ... 0.7 ms each and hipFree synchronises the device: the
// rocprof timeline of a device-resident 2048^2 run showed 42 % of its 0.19 s idle, most of it in front of the first kernel that
// follows an allocation (profiles/r04_driver_trace_before.txt).  Blocks are rounded up to an eighth of their leading power of two
// (<= 12.5 % slack), a freed block goes to the free list of its rounded size and serves the next request of that size.  Everything a
// context allocates is used on its one stream, so a recycled block needs no synchronisation: the new owner's first operation is
// ordered behind the old owner's last.  The cache is trimmed above `limit` bytes (default: a quarter of the device memory; env
// ICS_POOL_LIMIT_MB / debug switch pool_limit_mb, read when a context is created) and emptied when an allocation fails.
struct IcsPool {
  std::mutex mu;
  std::multimap<size_t, void*> free_;            // rounded size -> block
  std::unordered_map<void*, size_t> size_of;     // every block handed out or cached -> rounded size
  size_t cached = 0, limit = 0;
  static size_t round_up(size_t b) {
    if (b < 65536) b = 65536;
    size_t p2 = 65536;
    while (p2 * 2 <= b) p2 *= 2;                 // leading power of two
    const size_t q = p2 / 8;
    return (b + q - 1) / q * q;
  }
  void trim(size_t keep) {                       // (mu held) largest first
    while (cached > keep && !free_.empty()) {
      auto it = std::prev(free_.end());
      hipFree(it->second); size_of.erase(it->second); cached -= it->first; free_.erase(it);
    }
  }
  hipError_t alloc(void** p, size_t bytes) {
    const size_t r = round_up(bytes);
    std::lock_guard<std::mutex> g(mu);
    auto it = free_.find(r);
    if (it != free_.end()) { *p = it->second; cached -= r; free_.erase(it); return hipSuccess; }
    hipError_t e = hipMalloc(p, r);
    if (e != hipSuccess) { (void)hipGetLastError(); trim(0); e = hipMalloc(p, r); }
    if (e != hipSuccess) { (void)hipGetLastError(); *p = nullptr; return e; }
    size_of[*p] = r;
    return hipSuccess;
  }
  void release(void* p) {
    if (!p) return;
    std::lock_guard<std::mutex> g(mu);
    auto it = size_of.find(p);
    if (it == size_of.end()) { hipFree(p); return; }   // not ours
    free_.emplace(it->second, p); cached += it->second;
    if (cached > limit) trim(limit / 2);
  }
  void clear() { std::lock_guard<std::mutex> g(mu); trim(0); }
};

struct ics_ctx {
  IcsPool pool;
  int device;
  hipStream_t stream;
  int cus;
  char name[256];
  uint64_t hbm;
  void* scratch;            // device scratch of the standalone operators: grown on demand, kept between calls
  size_t scratch_bytes;
  hipEvent_t ev0, ev1;      // device time of the last standalone operator (kernels only, no transfers)
  float last_ms;
  // small pinned staging area for host -> device parameters of queued operations (the Gaussian weights of ics_img_resize): the copy
  // reads it asynchronously, `pin_ev` marks the last copy, the next writer waits for it (long done in practice) -- no stream
  // synchronisation per operation
  // second stream: the stop-test statistics of outer iteration i run here while the job's stream already works on iteration i + 1
  // (ics_rl_run, "overlap"; the events that order the two streams belong to the job)
  hipStream_t stream2 = nullptr;
  double* pin = nullptr;
  hipEvent_t pin_ev = nullptr;
  bool pin_used = false;
  static constexpr size_t PIN_DOUBLES = 8192;
};

// at least `bytes` of device scratch that persists between calls (no hipMalloc / hipFree per filter call)
static int ctx_scratch(ics_ctx* c, size_t bytes, void** p) {
  if (c->scratch_bytes < bytes) {
    if (c->scratch) { c->pool.release(c->scratch); c->scratch = nullptr; c->scratch_bytes = 0; }
    const size_t want = bytes + bytes / 4;
    hipError_t e = c->pool.alloc(&c->scratch, want);
    if (e != hipSuccess) { (void)hipGetLastError(); c->scratch = nullptr; return ICS_ENOMEM; }
    c->scratch_bytes = want;
  }
  *p = c->scratch;
  return ICS_OK;
}

struct ics_rl {
  ics_ctx* ctx;
  IcsGeom g;
  size_t frame_floats, origin;
  float *u, *u2, *ut, *gr, *f, *e;     // frame bases (origin = base + origin); u2 = ping-pong partner of u
  float* tvf;                           // TV term frame (tv_mode 1, allocated on first use)
  float* facc[2];                       // the image in accumulator order for 32-row / 64-row tiles (ics_image_acc.h), allocated on first use
  bool facc_valid[2];                   // ... and whether it still mirrors the image frame
  float *psf, *gradk, *wconv, *wcorr, *psf_caller, *partial;
  size_t partial_floats;                // size of `partial`
  float* psf_work;                      // PSF sizes above 63: working copy of k_psf (3*K*K floats), else NULL
  // overlap of the statistics with the next outer iteration: the reduction slots / DoF keys of outer iteration i are the set i & 1
  // (`par`; the stage API always uses set 0), the residual frame ping-pongs with e2, the PSF of the last finished iteration is kept
  int par;
  hipEvent_t ev_body[2], ev_stats[2];   // [i & 1]: iteration i's kernels are done / its scalars are on the host (created with e2)
  float* e2;                            // second residual frame (first overlapped run)
  float* psf_bak;                       // psf + psf_caller as they were when the running outer iteration started (blind, overlapped)
  double* gradk64;                      // row bands over several ranks: the gradient sums as float64 for the cross-rank all-reduce (first use)
  // PSF sizes 51 ... 255 on the matrix cores as nblk x nblk tap blocks of Kb x Kb (do_conv_blocks): weight tables of both
  // orientations, a scratch frame for the block results
  int blk_n, blk_kb;
  float *blk_conv, *blk_corr, *blk_scr, *blk_negf;   // blk_negf: -image (even block counts only: the chain of do_conv_blocks starts from it)
  bool negf_valid;
  uint32_t* blk_red;                    // reduction slots the block passes may scribble on (the maxima are taken over the sum)
  float *bt_conv, *bt_corr;  // Toeplitz fragment tables of the matrix-core convolution (MK <= 37), else NULL
  int gradk_blocks;
  int fused2_blocks;                    // persistent workgroups of the 32-row fused A11 + A13 kernel: three per CU (capped like gradk_blocks by the test switch)
  uint32_t* red;                        // INNER slots x ICS_RED_STRIDE keys
  uint32_t* dofkeys;                    // 4 words
  uint32_t* sched;                      // 16 words: tile-walk counters of the matrix-core convolutions (IcsConvArgs::sched)
  float* scal;                          // ICS_SC_COUNT
  double* dacc;                         // 8 accumulators of the window statistics
  uint32_t* ukey;                       // 2
  int* flags;                           // [0] frozen, [1] hasnan
  // stop-test scratch (allocated for the window of the last run)
  float2* z; float2* tw; float* weights;
  int P, logP, wt, wb, wl, wr;
  bool win_empty;
  float* h_scal;                        // pinned host mirror of scal (+ flags)
  bool uploaded;
  bool ut_is_u;                         // majoriser aliased to u (first inner iteration of an outer one, no copy made yet)
  // one hipGraph per outer iteration (small frames, use_graph): the launches of an outer iteration depend on which of the three
  // frames is u / ut / spare (the rotation of do_update has period 3), so up to three executables, valid for one parameter set
  struct Graph { float *u, *ut, *u2; hipGraphExec_t exec; };
  std::vector<Graph> graphs;
  ics_rl_params graph_sig;              // the parameters baked into the captured kernel arguments
  int graph_epoch;                      // ... and the state of the debug switches they were captured under
  // profiling
  std::vector<hipEvent_t> ev;
  struct EvPair { int b, e, cls; };     // a bracketed launch group: events ev[b] .. ev[e]
  std::vector<EvPair> ev_pairs;
  size_t ev_used;
  int ev_open = -1, ev_open_cls = 0;    // begin() without its end() yet
  int ev_chain = -1;                    // the event the last end() recorded, while nothing else has been queued behind it (Prof::begin)
  hipStream_t ev_chain_stream = nullptr;
  hipEvent_t ev_begin, ev_end;
  // FFT-tile pipeline (round 5; ics_conv_fft.hip, ics_planar.hip): channel-planar mirrors of the frames (ics_common.h), allocated by the
  // first run that uses it.  A mirror belongs to a BUFFER, not to a role: u / ut / u2 and e / e2 rotate as pointers, the table is looked
  // up by the HWC pointer's value.  fft_on = a run / stage on the pipeline is in progress: the do_* helpers work on the mirrors and
  // pack_weights also builds the two spectra.
  struct Twin { float* hwc; float* pl; };
  Twin twins[8];
  int ntwins;
  float *spec_conv, *spec_corr;
  float* fspec;         // mode 2 of the tile convolutions (A1 + A3 in one unit): the image windows' spectra, valid while fspec_valid
  bool fspec_valid;
  bool conv2_off;       // the spectra did not fit the device memory once: this job runs A1 and A3 as two kernels from then on
  bool fft_on;
  bool plf_valid;                       // the mirror of the image frame still mirrors it (every writer of j->f calls image_changed)
  // small frames (ics_small.hip): the inner iterations of an outer one as a cooperative launch
  float* small_part;                    // the tiles' shares of the PSF gradient (3 x tiles x K^2)
  unsigned long long* small_bar;        // the grid barrier's counters, zeroed at the start of a run
  unsigned long long* small_keys;       // the tiles' step-size maxima (8 x workgroups)
  unsigned long long small_gen;         // barriers passed since then
  bool small_off;                       // the cooperative launch was refused once: this job stays on the multi-launch path
  bool small_bak;                       // the next cooperative launch first copies psf / psf_caller to psf_bak (overlapped statistics, blind)
};

static inline float* org(ics_rl* j, float* base) { return base + j->origin; }
static inline float* pl_of(ics_rl* j, const float* hwc) {
  for (int i = 0; i < j->ntwins; ++i) if (j->twins[i].hwc == hwc) return j->twins[i].pl;
  return nullptr;
}
// origin of the planar mirror of an HWC frame buffer
static inline float* porg(ics_rl* j, const float* hwc) { float* p = pl_of(j, hwc); return p ? p + ics_planar_origin(j->g) : nullptr; }
// majoriser frame: pyx:462 `ut = u.copy()` is realised without a copy -- until the first update of the outer
// iteration ut IS u; that update writes out of place and the old u frame becomes ut (buffer rotation)
static inline float* ut_of(ics_rl* j) { return j->ut_is_u ? j->u : j->ut; }
static inline uint32_t* red_of(ics_rl* j) { return j->red + (size_t)j->par * 8 * ICS_RED_STRIDE; }
static inline uint32_t* dof_of(ics_rl* j) { return j->dofkeys + 4 * j->par; }

// -------------------------------------------------------------------------------------------------
extern "C" int ics_abi_version(void) { return ICS_ABI_VERSION; }
extern "C" size_t ics_rl_params_size(void) { return sizeof(ics_rl_params); }
extern "C" size_t ics_rl_stats_size(void) { return sizeof(ics_rl_stats); }

// test / measurement switches (ics_common.h IcsDebug): exported, deliberately absent from include/ics_hip.h
static std::atomic<int> g_debug_epoch{0};   // bumped by every ics_debug_set: captured graphs carry the switches' effects in their kernel arguments
extern "C" int ics_debug_set(const char* name, int value) {
  if (!name) return -1;
  IcsDebug& d = ics_debug();
  struct { const char* n; std::atomic<int>* v; } tab[] = {
      {"max_wgs", &d.max_wgs}, {"dynamic_tiles", &d.dynamic_tiles}, {"conv_rs", &d.conv_rs}, {"conv_nh", &d.conv_nh}, {"conv_path", &d.conv_path},
      {"fused_gradk", &d.fused_gradk}, {"update_wg_per_cu", &d.update_wg_per_cu}, {"update_kernel", &d.update_kernel}, {"fused_rs", &d.fused_rs},
      {"planar_image", &d.planar_image}, {"pam_exact", &d.pam_exact}, {"fail_window_alloc", &d.fail_window_alloc}, {"graph", &d.graph}, {"pool_limit_mb", &d.pool_limit_mb}, {"overlap", &d.overlap}, {"fft_gradk", &d.fft_gradk}, {"fft_fused", &d.fft_fused}, {"fft_conv2", &d.fft_conv2}, {"fft_rot", &d.fft_rot}, {"small_iter", &d.small_iter}, {"small_trace", &d.small_trace}, {"fail_small_launch", &d.fail_small_launch}};
  for (auto& t : tab)
    if (strcmp(t.n, name) == 0) { t.v->store(value, std::memory_order_relaxed); g_debug_epoch.fetch_add(1, std::memory_order_relaxed); return 0; }
  return -1;
}
extern "C" int ics_debug_get(const char* name, int* value) {
  if (!name || !value) return -1;
  IcsDebug& d = ics_debug();
  struct { const char* n; std::atomic<int>* v; } tab[] = {
      {"max_wgs", &d.max_wgs}, {"dynamic_tiles", &d.dynamic_tiles}, {"conv_rs", &d.conv_rs}, {"conv_nh", &d.conv_nh}, {"conv_path", &d.conv_path},
      {"fused_gradk", &d.fused_gradk}, {"update_wg_per_cu", &d.update_wg_per_cu}, {"update_kernel", &d.update_kernel}, {"fused_rs", &d.fused_rs},
      {"planar_image", &d.planar_image}, {"pam_exact", &d.pam_exact}, {"fail_window_alloc", &d.fail_window_alloc}, {"graph", &d.graph}, {"pool_limit_mb", &d.pool_limit_mb}, {"overlap", &d.overlap}, {"fft_gradk", &d.fft_gradk}, {"fft_fused", &d.fft_fused}, {"fft_conv2", &d.fft_conv2}, {"fft_rot", &d.fft_rot}, {"small_iter", &d.small_iter}, {"small_trace", &d.small_trace}, {"fail_small_launch", &d.fail_small_launch}};
  for (auto& t : tab)
    if (strcmp(t.n, name) == 0) { *value = t.v->load(std::memory_order_relaxed); return 0; }
  return -1;
}
extern "C" const char* ics_last_error(void) { return g_err; }

extern "C" int ics_device_count(int* count) {
  if (!count) return fail(ICS_EINVAL, "count is NULL");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) { *count = 0; return fail(ICS_ENODEV, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
  *count = n;
  return ICS_OK;
}

extern "C" int ics_ctx_create(int device, ics_ctx** out) {
  if (!out) return fail(ICS_EINVAL, "out is NULL");
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return fail(ICS_ENODEV, "no HIP device available (%s); libics_hip has no CPU fallback", e == hipSuccess ? "count = 0" : hipGetErrorString(e));
  if (device < 0 || device >= n) return fail(ICS_EINVAL, "device %d out of range (0..%d)", device, n - 1);
  HIPCHK(hipSetDevice(device));
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(ICS_ENODEV, "device %d is %s; this library only contains gfx950 (MI355X) code", device, prop.gcnArchName);
  ics_ctx* c = new ics_ctx();
  c->device = device;
  c->scratch = nullptr; c->scratch_bytes = 0; c->last_ms = 0.f;
  {
    const int lim = ics_debug().pool_limit_mb.load(std::memory_order_relaxed);   // ICS_POOL_LIMIT_MB, read once per process
    c->pool.limit = lim >= 0 ? (size_t)lim << 20 : (size_t)prop.totalGlobalMem / 4;
  }
  hipEventCreate(&c->ev0); hipEventCreate(&c->ev1);
  c->cus = prop.multiProcessorCount;
  c->hbm = prop.totalGlobalMem;
  snprintf(c->name, sizeof c->name, "%s (%s)", prop.name, prop.gcnArchName);
  hipError_t se = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  if (se != hipSuccess) { delete c; return fail(ICS_EHIP, "hipStreamCreate: %s", hipGetErrorString(se)); }
  *out = c;
  return ICS_OK;
}

extern "C" void ics_ctx_destroy(ics_ctx* c) {
  if (!c) return;
  hipSetDevice(c->device);
  hipStreamSynchronize(c->stream);
  if (c->scratch) c->pool.release(c->scratch);
  c->pool.clear();
  if (c->pin) hipHostFree(c->pin);
  if (c->pin_ev) hipEventDestroy(c->pin_ev);
  if (c->stream2) { hipStreamSynchronize(c->stream2); hipStreamDestroy(c->stream2); }
  hipEventDestroy(c->ev0); hipEventDestroy(c->ev1);
  hipStreamDestroy(c->stream);
  delete c;
}

extern "C" int ics_ctx_synchronize(ics_ctx* c) {
  if (!c) return fail(ICS_EINVAL, "ctx is NULL");
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipStreamSynchronize(c->stream));
  return ICS_OK;
}

extern "C" int ics_ctx_last_kernel_ms(ics_ctx* c, float* ms) {
  if (!c || !ms) return fail(ICS_EINVAL, "NULL argument");
  *ms = c->last_ms;
  return ICS_OK;
}

extern "C" int ics_ctx_info(ics_ctx* c, char* name, size_t name_len, int* cus, uint64_t* hbm) {
  if (!c) return fail(ICS_EINVAL, "ctx is NULL");
  if (name && name_len) { strncpy(name, c->name, name_len - 1); name[name_len - 1] = 0; }
  if (cus) *cus = c->cus;
  if (hbm) *hbm = c->hbm;
  return ICS_OK;
}

// -------------------------------------------------------------------------------------------------
// PSF sizes 129 ... ICS_PSF_MAX: only as tap blocks on the matrix cores -- convolutions as blocks of <= 33 x 33 taps (do_conv_blocks), the
// gradient as blocks of <= 31 x 31 (do_gradk_split); the run-time-sized fp32 kernels of ics_big.hip (ICS_CONV_VECTOR) stop at 127.
#define ICS_PSF_MAX 255
static bool psf_blocks_only(int K) { return K > 127 && K <= ICS_PSF_MAX && (K & 1); }
static bool psf_supported(int K) { return ics_conv_supported(K) || ics_big_supported(K) || psf_blocks_only(K); }

// Device allocation, zero-filled ON THE GIVEN STREAM: the job's stream is non-blocking, so a
// null-stream hipMemset would not be ordered with the uploads/kernels that follow on it.
template <typename T>
static int dalloc(ics_ctx* c, T** p, size_t count, bool zero = true) {
  *p = nullptr;
  if (hipError_t e = c->pool.alloc((void**)p, count * sizeof(T)); e != hipSuccess)
    return fail(ICS_ENOMEM, "device allocation of %zu bytes failed: %s", count * sizeof(T), hipGetErrorString(e));
  if (zero) HIPCHK(hipMemsetAsync(*p, 0, count * sizeof(T), c->stream));
  return ICS_OK;
}

// ... or collected in `zl` and zero-filled by ONE launch (flush_zero): a new job's ~24 buffers as 24 memsets cost the host ~35 us each
// (deblur_module creates a job per pyramid level and phase: 8 % of a resident 2048^2 run were the gaps in front of those fills)
struct ZeroList { std::vector<std::pair<void*, size_t>> items; };
template <typename T>
static int dalloc(ics_ctx* c, T** p, size_t count, ZeroList* zl) {
  const int rc = dalloc(c, p, count, false);
  if (rc == ICS_OK) zl->items.emplace_back((void*)*p, count * sizeof(T));
  return rc;
}
static int flush_zero(ics_ctx* c, ZeroList& zl) {
  size_t i = 0;
  while (i < zl.items.size()) {
    IcsZeroArgs a{};
    unsigned long long run = 0;
    for (; a.count < ICS_ZERO_MAX && i < zl.items.size(); ++i) {
      a.p[a.count] = zl.items[i].first;
      run += (zl.items[i].second + 15) / 16;        // (pool blocks are multiples of 8 KiB: the rounding stays inside the block)
      a.end16[a.count++] = run;
    }
    HIPCHK(ics_launch_zero_many(a, c->stream));
  }
  zl.items.clear();
  return ICS_OK;
}

extern "C" void ics_rl_destroy(ics_rl* j) {
  if (!j) return;
  hipSetDevice(j->ctx->device);
  hipStreamSynchronize(j->ctx->stream);
  void* ptrs[] = {j->facc[0], j->facc[1], j->tvf, j->u, j->u2, j->ut, j->gr, j->f, j->e, j->psf, j->gradk, j->wconv, j->wcorr, j->bt_conv, j->bt_corr, j->psf_caller, j->partial, j->psf_work, j->blk_conv, j->blk_corr, j->blk_scr, j->blk_negf, j->blk_red,
                  j->red, j->dofkeys, j->sched, j->scal, j->dacc, j->ukey, j->flags, j->z, j->tw, j->weights, j->gradk64, j->e2, j->psf_bak, j->small_part, j->small_bar, j->small_keys};
  if (j->ctx->stream2) hipStreamSynchronize(j->ctx->stream2);   // (the statistics' stream uses the job's buffers as well: drain it before they are recycled)
  for (void* p : ptrs) if (p) j->ctx->pool.release(p);   // (recycled by the context: ordered on its stream, no hipFree synchronisation)
  for (int i = 0; i < j->ntwins; ++i) if (j->twins[i].pl) j->ctx->pool.release(j->twins[i].pl);
  if (j->spec_conv) j->ctx->pool.release(j->spec_conv);
  if (j->spec_corr) j->ctx->pool.release(j->spec_corr);
  if (j->fspec) j->ctx->pool.release(j->fspec);
  for (auto& g : j->graphs) hipGraphExecDestroy(g.exec);
  for (int i = 0; i < 2; ++i) { if (j->ev_body[i]) hipEventDestroy(j->ev_body[i]); if (j->ev_stats[i]) hipEventDestroy(j->ev_stats[i]); }
  if (j->h_scal) hipHostFree(j->h_scal);
  for (hipEvent_t e : j->ev) hipEventDestroy(e);
  if (j->ev_begin) hipEventDestroy(j->ev_begin);
  if (j->ev_end) hipEventDestroy(j->ev_end);
  delete j;
}

extern "C" int ics_rl_create(ics_ctx* c, int M, int N, int MK, ics_rl** out) {
  if (!c || !out) return fail(ICS_EINVAL, "NULL argument");
  *out = nullptr;
  if (M < 1 || N < 1) return fail(ICS_EINVAL, "image size %dx%d", M, N);
  if (MK < 3 || !(MK & 1)) return fail(ICS_EINVAL, "MK must be odd and >= 3 (got %d)", MK);
  if (!psf_supported(MK)) return fail(ICS_ENOSUP, "PSF size %d not supported (odd sizes 3..%d)", MK, ICS_PSF_MAX);
  HIPCHK(hipSetDevice(c->device));
  ics_rl* j = new ics_rl();  // value-initialised: every pointer/flag starts at 0
  j->ctx = c;
  j->g = ics_make_geom(M, N, MK);
  j->frame_floats = ics_frame_floats(j->g);
  j->origin = ics_origin_offset(j->g);
  // the matrix-core kernels address a frame through a raw buffer descriptor with 32-bit byte offsets
  if (j->frame_floats * 4 >= (size_t)ICS_FRAME_LIMIT_BYTES) {
    delete j;
    return fail(ICS_ENOSUP, "a %d x %d frame with a %d x %d PSF takes %.2f GB; frames are limited to 2 GiB (about 13000 x 13000 px)", M, N, MK, MK,
                (double)ics_frame_floats(ics_make_geom(M, N, MK)) * 4e-9);
  }
  const size_t n = (size_t)3 * MK * MK;
  const int nt = 16 * ((MK + 15) / 16);
  j->gradk_blocks = ics_gradk_blocks(j->g, c->cus);
  j->fused2_blocks = 3 * c->cus;
  if (const int mw = ics_debug().max_wgs.load(std::memory_order_relaxed); mw > 0 && j->fused2_blocks > mw) j->fused2_blocks = mw;
  int rc;
#define TRY(x) if ((rc = (x)) != ICS_OK) { ics_rl_destroy(j); return rc; }
  hipStream_t s = c->stream;
  ZeroList zl;
  TRY(dalloc(c, &j->u, j->frame_floats, &zl)); TRY(dalloc(c, &j->u2, j->frame_floats, &zl)); TRY(dalloc(c, &j->ut, j->frame_floats, &zl)); TRY(dalloc(c, &j->gr, j->frame_floats, &zl));
  TRY(dalloc(c, &j->f, j->frame_floats, &zl)); TRY(dalloc(c, &j->e, j->frame_floats, &zl));
  TRY(dalloc(c, &j->psf, n, &zl)); TRY(dalloc(c, &j->gradk, n, &zl)); TRY(dalloc(c, &j->psf_caller, n, &zl));
  if (MK > 63) TRY(dalloc(c, &j->psf_work, n, &zl));   // (k_psf<BIG>)
  if (MK >= 51) {   // tap blocks: the fewest blocks of a size the matrix-core convolution is built for (odd, <= 33)
    j->blk_n = (MK + 32) / 33;
    j->blk_kb = ((MK + j->blk_n - 1) / j->blk_n) | 1;
    const size_t tf = ics_conv_mfma_table_floats(j->blk_kb);
    TRY(dalloc(c, &j->blk_conv, tf * j->blk_n * j->blk_n, &zl)); TRY(dalloc(c, &j->blk_corr, tf * j->blk_n * j->blk_n, &zl));
    TRY(dalloc(c, &j->blk_scr, j->frame_floats, &zl));
    if (!(j->blk_n & 1)) TRY(dalloc(c, &j->blk_negf, j->frame_floats, &zl));
    TRY(dalloc(c, &j->blk_red, (size_t)ICS_RED_STRIDE, &zl));
  }
  TRY(dalloc(c, &j->wconv, (size_t)(MK + 1) * j->g.wrow, &zl)); TRY(dalloc(c, &j->wcorr, (size_t)(MK + 1) * j->g.wrow, &zl));
  if (ics_conv_mfma_supported(MK)) { TRY(dalloc(c, &j->bt_conv, ics_conv_mfma_table_floats(MK), &zl)); TRY(dalloc(c, &j->bt_corr, ics_conv_mfma_table_floats(MK), &zl)); }
  // (129 ...: the gradient only ever runs as 31 x 31 blocks -- 2 * CUs workgroups of 3 x 32 x 32 partial sums each, do_gradk_split)
  j->partial_floats = psf_blocks_only(MK) ? (size_t)2 * c->cus * 3 * 32 * 32
                                          : (size_t)(j->gradk_blocks > j->fused2_blocks ? j->gradk_blocks : j->fused2_blocks) * 3 * nt * nt;
  TRY(dalloc(c, &j->partial, j->partial_floats, &zl));
  TRY(dalloc(c, &j->red, (size_t)2 * 8 * ICS_RED_STRIDE, &zl)); TRY(dalloc(c, &j->dofkeys, (size_t)2 * 4, &zl)); TRY(dalloc(c, &j->sched, (size_t)16, &zl));   // (two sets: ics_rl::par)
  TRY(dalloc(c, &j->scal, (size_t)ICS_SC_COUNT, &zl)); TRY(dalloc(c, &j->dacc, (size_t)8, &zl)); TRY(dalloc(c, &j->ukey, (size_t)2, &zl)); TRY(dalloc(c, &j->flags, (size_t)4, &zl));
  TRY(flush_zero(c, zl));
#undef TRY
  hipError_t e = hipHostMalloc((void**)&j->h_scal, 2 * (ICS_SC_COUNT + 4) * sizeof(float), hipHostMallocDefault);   // (one mirror per set)
  if (e != hipSuccess) { ics_rl_destroy(j); return fail(ICS_ENOMEM, "hipHostMalloc: %s", hipGetErrorString(e)); }
  hipEventCreate(&j->ev_begin); hipEventCreate(&j->ev_end);
  e = hipStreamSynchronize(s);
  if (e != hipSuccess) { ics_rl_destroy(j); return fail(ICS_EHIP, "hipStreamSynchronize: %s", hipGetErrorString(e)); }
  *out = j;
  return ICS_OK;
}

static int copy_in(ics_rl* j, float* frame, const float* host, int rows, int cols_px, int oy, int ox) {
  float* dst = org(j, frame) + (ptrdiff_t)oy * j->g.pitch + 3 * ox;
  HIPCHK(hipMemcpy2DAsync(dst, (size_t)j->g.pitch * 4, host, (size_t)cols_px * 12, (size_t)cols_px * 12, rows,
                          hipMemcpyHostToDevice, j->ctx->stream));
  return ICS_OK;
}
static int copy_out(ics_rl* j, float* frame, float* host, int rows, int cols_px, int oy, int ox) {
  const float* src = org(j, frame) + (ptrdiff_t)oy * j->g.pitch + 3 * ox;
  HIPCHK(hipMemcpy2DAsync(host, (size_t)cols_px * 12, src, (size_t)j->g.pitch * 4, (size_t)cols_px * 12, rows,
                          hipMemcpyDeviceToHost, j->ctx->stream));
  return ICS_OK;
}

// the accumulator-order copies of the image follow the image frame: every writer of j->f calls this
static inline void image_changed(ics_rl* j) { j->facc_valid[0] = j->facc_valid[1] = false; j->negf_valid = false; j->plf_valid = false; j->fspec_valid = false; }

// (re)build the accumulator-order image for tile height 16 * RS if it is missing or stale; queued on the job's stream
static int ensure_image_acc(ics_rl* j, int RS) {
  const int k = RS == 2 ? 0 : 1;
  if (!j->facc[k]) {
    hipError_t e = j->ctx->pool.alloc((void**)&j->facc[k], ics_image_acc_floats(j->g, RS) * sizeof(float));
    if (e != hipSuccess) { (void)hipGetLastError(); j->facc[k] = nullptr; return fail(ICS_ENOMEM, "accumulator-order image: %s", hipGetErrorString(e)); }
    j->facc_valid[k] = false;
  }
  if (!j->facc_valid[k]) {
    hipError_t e = ics_launch_image_acc(j->f + j->origin, j->g, RS, j->facc[k], j->ctx->stream);
    if (e != hipSuccess) return fail(ICS_EHIP, "k_image_acc: %s", hipGetErrorString(e));
    j->facc_valid[k] = true;
  }
  return ICS_OK;
}

static int pack_weights(ics_rl* j, int do_step, float step, int correlation, hipStream_t s) {
  IcsPsfArgs a;
  a.psf = j->psf; a.gradk = j->gradk; a.wconv = j->wconv; a.wcorr = j->wcorr; a.bt_conv = j->bt_conv; a.bt_corr = j->bt_corr; a.psf_caller = j->psf_caller; a.work = j->psf_work;
  a.scal = j->scal; a.frozen = j->flags; a.step = step; a.K = j->g.K; a.wrow = j->g.wrow;
  a.correlation = correlation; a.do_step = do_step;
  HIPCHK(ics_launch_psf(a, s));
  if (j->blk_conv) HIPCHK(ics_launch_pack_blocks(j->psf, j->g.K, j->blk_kb, j->blk_n, j->blk_conv, j->blk_corr, ics_conv_mfma_table_floats(j->blk_kb), s));
  if (j->fft_on) {   // conj(DFT2(W)) / 128^2 of both orientations (PSF sizes above 97: of every tap block)
    int nb = 0, kb = 0;
    if (ics_conv_fft_blk_supported(j->g.K)) ics_conv_fft_blk_shape(j->g.K, &nb, &kb);
    HIPCHK(ics_launch_fft_spectrum(j->psf, j->g.K, j->spec_conv, j->spec_corr, s, nb, kb));
  }
  return ICS_OK;
}

// ---- FFT-tile pipeline: mirrors ---------------------------------------------------------------------------------------------------------
// every HWC frame buffer the pipeline touches gets a planar mirror (zero-filled: the aprons of a mirror are never written either)
static int ensure_planar(ics_rl* j) {
  float* want[] = {j->u, j->u2, j->ut, j->gr, j->f, j->e, j->e2, j->tvf};
  for (float* h : want) {
    if (!h || pl_of(j, h)) continue;
    if (j->ntwins >= 8) return fail(ICS_ESTATE, "planar mirror table full");
    float* pl = nullptr;
    RC(dalloc(j->ctx, &pl, ics_planar_floats(j->g)));
    j->twins[j->ntwins].hwc = h; j->twins[j->ntwins].pl = pl; ++j->ntwins;
    if (h == j->f) j->plf_valid = false;
  }
  int nb = 1, kb = 0;
  if (ics_conv_fft_blk_supported(j->g.K)) ics_conv_fft_blk_shape(j->g.K, &nb, &kb);      // (one spectrum per tap block and orientation)
  if (!j->spec_conv) RC(dalloc(j->ctx, &j->spec_conv, (size_t)nb * nb * ics_conv_fft_spectrum_floats()));
  if (!j->spec_corr) RC(dalloc(j->ctx, &j->spec_corr, (size_t)nb * nb * ics_conv_fft_spectrum_floats()));
  return ICS_OK;
}
// whole-buffer copies HWC -> mirror / mirror -> HWC (run and stage boundaries), and the stop-test window mirror -> HWC
static int to_planar(ics_rl* j, float* hwc, hipStream_t s) {
  HIPCHK(ics_launch_planar_convert(true, hwc, pl_of(j, hwc), j->g, true, 0, 0, 0, 0, s));
  return ICS_OK;
}
static int from_planar(ics_rl* j, float* hwc, hipStream_t s) {   // the u-frame only: the aprons of both stay zero
  HIPCHK(ics_launch_planar_convert(false, pl_of(j, hwc), hwc, j->g, false, 0, j->g.uM, 0, j->g.uN, s));
  return ICS_OK;
}
struct FftScope {   // fft_on for the duration of a run / stage, whatever path leaves it
  ics_rl* j;
  bool back = false;   // ics_rl_run: the mirrors hold newer data than the HWC frames (set once the run has started, cleared by the regular conversion)
  // an error return in the middle of a run: bring u and the residual back from the mirrors as well as the device still allows, so that a
  // later ics_rl_read / stage sees the state the failed run left, not a rotated buffer with stale HWC contents
  ~FftScope() {
    if (!j) return;
    if (j->fft_on && back) {
      if (pl_of(j, j->u)) (void)from_planar(j, j->u, j->ctx->stream);
      if (pl_of(j, j->e)) (void)from_planar(j, j->e, j->ctx->stream);
      (void)hipStreamSynchronize(j->ctx->stream); (void)hipGetLastError();
    }
    j->fft_on = false;
  }
};

extern "C" int ics_rl_upload(ics_rl* j, const float* image, const float* u, const float* psf) {
  if (!j) return fail(ICS_EINVAL, "job is NULL");
  HIPCHK(hipSetDevice(j->ctx->device));
  hipStream_t s = j->ctx->stream;
  const IcsGeom& g = j->g;
  int rc;
  if (image) image_changed(j);
  if (image && (rc = copy_in(j, j->f, image, g.M, g.N, g.pad, g.pad)) != ICS_OK) return rc;
  if (u && (rc = copy_in(j, j->u, u, g.uM, g.uN, 0, 0)) != ICS_OK) return rc;
  if (psf) {
    const size_t n = (size_t)3 * g.K * g.K * 4;
    HIPCHK(hipMemcpyAsync(j->psf, psf, n, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(j->psf_caller, psf, n, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemsetAsync(j->flags, 0, 4 * sizeof(int), s));
    if ((rc = pack_weights(j, 0, 0.f, 0, s)) != ICS_OK) return rc;
  }
  HIPCHK(hipStreamSynchronize(s));
  if (image && u && psf) j->uploaded = true;
  return ICS_OK;
}

extern "C" int ics_rl_download(ics_rl* j, float* u, float* psf_local, float* psf_caller) {
  if (!j) return fail(ICS_EINVAL, "job is NULL");
  HIPCHK(hipSetDevice(j->ctx->device));
  hipStream_t s = j->ctx->stream;
  const IcsGeom& g = j->g;
  int rc;
  if (u && (rc = copy_out(j, j->u, u, g.uM, g.uN, 0, 0)) != ICS_OK) return rc;
  const size_t n = (size_t)3 * g.K * g.K * 4;
  if (psf_local) HIPCHK(hipMemcpyAsync(psf_local, j->psf, n, hipMemcpyDeviceToHost, s));
  if (psf_caller) HIPCHK(hipMemcpyAsync(psf_caller, j->psf_caller, n, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  return ICS_OK;
}

static int frame_of(ics_rl* j, int which, float** frame, int* rows, int* cols, int* oy, int* ox) {
  const IcsGeom& g = j->g;
  switch (which) {
    case ICS_BUF_U: *frame = j->u; break;
    case ICS_BUF_UT: *frame = j->ut; break;
    case ICS_BUF_GRADU: *frame = j->gr; break;
    case ICS_BUF_IMAGE: *frame = j->f; break;
    case ICS_BUF_ERROR: *frame = j->e; break;
    case ICS_BUF_TV: if (!j->tvf) return -1; *frame = j->tvf; break;
    default: return -1;
  }
  if (which == ICS_BUF_IMAGE || which == ICS_BUF_ERROR) { *rows = g.M; *cols = g.N; *oy = g.pad; *ox = g.pad; }
  else { *rows = g.uM; *cols = g.uN; *oy = 0; *ox = 0; }
  return 0;
}

extern "C" int ics_rl_read(ics_rl* j, int which, float* host, size_t count) {
  if (!j || !host) return fail(ICS_EINVAL, "NULL argument");
  HIPCHK(hipSetDevice(j->ctx->device));
  hipStream_t s = j->ctx->stream;
  const size_t n = (size_t)3 * j->g.K * j->g.K;
  float* frame; int rows, cols, oy, ox;
  if (frame_of(j, which, &frame, &rows, &cols, &oy, &ox) == 0) {
    if (count != (size_t)rows * cols * 3) return fail(ICS_EINVAL, "buffer %d holds %zu floats, got %zu", which, (size_t)rows * cols * 3, count);
    int rc = copy_out(j, frame, host, rows, cols, oy, ox);
    if (rc != ICS_OK) return rc;
  } else if (which == ICS_BUF_PSF || which == ICS_BUF_GRADK) {
    if (count != n) return fail(ICS_EINVAL, "buffer %d holds %zu floats, got %zu", which, n, count);
    HIPCHK(hipMemcpyAsync(host, which == ICS_BUF_PSF ? j->psf : j->gradk, n * 4, hipMemcpyDeviceToHost, s));
  } else if (which == ICS_BUF_SCALARS) {
    if (count != ICS_SC_COUNT) return fail(ICS_EINVAL, "scalars hold %d floats", ICS_SC_COUNT);
    HIPCHK(hipMemcpyAsync(host, j->scal, ICS_SC_COUNT * 4, hipMemcpyDeviceToHost, s));
  } else if (which == ICS_BUF_RED) {
    if (count != ICS_RED_STRIDE) return fail(ICS_EINVAL, "the reduction slot holds %d words", ICS_RED_STRIDE);
    HIPCHK(hipMemcpyAsync(host, j->red, 12 * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(host + 12, j->dofkeys, 4 * 4, hipMemcpyDeviceToHost, s));   // [12] min key, [13] max key, [14] NaN flag of the DoF mask
  } else {
    return fail(ICS_EINVAL, "unknown buffer %d", which);
  }
  HIPCHK(hipStreamSynchronize(s));
  return ICS_OK;
}

extern "C" int ics_rl_write(ics_rl* j, int which, const float* host, size_t count) {
  if (!j || !host) return fail(ICS_EINVAL, "NULL argument");
  HIPCHK(hipSetDevice(j->ctx->device));
  hipStream_t s = j->ctx->stream;
  const size_t n = (size_t)3 * j->g.K * j->g.K;
  float* frame; int rows, cols, oy, ox;
  if (frame_of(j, which, &frame, &rows, &cols, &oy, &ox) == 0) {
    if (count != (size_t)rows * cols * 3) return fail(ICS_EINVAL, "buffer %d holds %zu floats, got %zu", which, (size_t)rows * cols * 3, count);
    if (which == ICS_BUF_IMAGE) image_changed(j);
    int rc = copy_in(j, frame, host, rows, cols, oy, ox);
    if (rc != ICS_OK) return rc;
  } else if (which == ICS_BUF_PSF || which == ICS_BUF_GRADK) {
    if (count != n) return fail(ICS_EINVAL, "buffer %d holds %zu floats, got %zu", which, n, count);
    HIPCHK(hipMemcpyAsync(which == ICS_BUF_PSF ? j->psf : j->gradk, host, n * 4, hipMemcpyHostToDevice, s));
    if (which == ICS_BUF_PSF) { int rc = pack_weights(j, 0, 0.f, 0, s); if (rc != ICS_OK) return rc; }
  } else if (which == ICS_BUF_RED) {
    if (count != ICS_RED_STRIDE) return fail(ICS_EINVAL, "the reduction slot holds %d words", ICS_RED_STRIDE);
    HIPCHK(hipMemcpyAsync(j->red, host, ICS_RED_STRIDE * 4, hipMemcpyHostToDevice, s));
  } else {
    return fail(ICS_EINVAL, "buffer %d is not writable", which);
  }
  HIPCHK(hipStreamSynchronize(s));
  return ICS_OK;
}

static int rows_io(ics_rl* j, int which, int row0, int nrows, float* host, bool to_host) {
  if (!j || !host) return fail(ICS_EINVAL, "NULL argument");
  HIPCHK(hipSetDevice(j->ctx->device));
  float* frame; int rows, cols, oy, ox;
  if (frame_of(j, which, &frame, &rows, &cols, &oy, &ox) != 0) return fail(ICS_EINVAL, "buffer %d is not a frame", which);
  if (row0 < 0 || nrows < 1 || row0 + nrows > rows) return fail(ICS_EINVAL, "rows [%d, %d) outside the %d rows of buffer %d", row0, row0 + nrows, rows, which);
  float* dev = org(j, frame) + (ptrdiff_t)(oy + row0) * j->g.pitch + 3 * ox;
  if (!to_host && which == ICS_BUF_IMAGE) image_changed(j);
  if (to_host) HIPCHK(hipMemcpy2DAsync(host, (size_t)cols * 12, dev, (size_t)j->g.pitch * 4, (size_t)cols * 12, nrows, hipMemcpyDeviceToHost, j->ctx->stream));
  else HIPCHK(hipMemcpy2DAsync(dev, (size_t)j->g.pitch * 4, host, (size_t)cols * 12, (size_t)cols * 12, nrows, hipMemcpyHostToDevice, j->ctx->stream));
  HIPCHK(hipStreamSynchronize(j->ctx->stream));
  return ICS_OK;
}
extern "C" int ics_rl_read_rows(ics_rl* j, int which, int row0, int nrows, float* host) { return rows_io(j, which, row0, nrows, host, true); }
extern "C" int ics_rl_write_rows(ics_rl* j, int which, int row0, int nrows, const float* host) { return rows_io(j, which, row0, nrows, const_cast<float*>(host), false); }

// device-to-device rows between two jobs (lib/banded.py: halo exchange, stop-test gather).  Ordering: the source stream is
// drained, the copy runs on the destination stream and is waited for -- the band driver is host-synchronous per stage anyway.
extern "C" int ics_rl_copy_rows(ics_rl* dst, int dst_which, int dst_row0, ics_rl* src, int src_which, int src_row0, int nrows) {
  if (!dst || !src) return fail(ICS_EINVAL, "NULL argument");
  float *df, *sf; int drows, dcols, doy, dox, srows, scols, soy, sox;
  if (frame_of(dst, dst_which, &df, &drows, &dcols, &doy, &dox) != 0) return fail(ICS_EINVAL, "buffer %d is not a frame", dst_which);
  if (frame_of(src, src_which, &sf, &srows, &scols, &soy, &sox) != 0) return fail(ICS_EINVAL, "buffer %d is not a frame", src_which);
  if (dcols != scols) return fail(ICS_EINVAL, "row length %d (destination) != %d (source)", dcols, scols);
  if (nrows < 1 || dst_row0 < 0 || dst_row0 + nrows > drows || src_row0 < 0 || src_row0 + nrows > srows)
    return fail(ICS_EINVAL, "rows [%d, %d) of %d <- rows [%d, %d) of %d", dst_row0, dst_row0 + nrows, drows, src_row0, src_row0 + nrows, srows);
  if (dst_which == ICS_BUF_IMAGE) image_changed(dst);
  const int dd = dst->ctx->device, sd = src->ctx->device;
  if (dd != sd) {
    int can = 0;
    HIPCHK(hipDeviceCanAccessPeer(&can, dd, sd));
    if (!can) return fail(ICS_ENOSUP, "device %d cannot access device %d: no peer path", dd, sd);
    HIPCHK(hipSetDevice(dd));
    hipError_t pe = hipDeviceEnablePeerAccess(sd, 0);
    if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) return fail(ICS_EHIP, "hipDeviceEnablePeerAccess(%d): %s", sd, hipGetErrorString(pe));
    (void)hipGetLastError();
  }
  HIPCHK(hipSetDevice(sd));
  HIPCHK(hipStreamSynchronize(src->ctx->stream));
  HIPCHK(hipSetDevice(dd));
  const float* sp = org(src, sf) + (ptrdiff_t)(soy + src_row0) * src->g.pitch + 3 * sox;
  float* dp = org(dst, df) + (ptrdiff_t)(doy + dst_row0) * dst->g.pitch + 3 * dox;
  HIPCHK(hipMemcpy2DAsync(dp, (size_t)dst->g.pitch * 4, sp, (size_t)src->g.pitch * 4, (size_t)dcols * 12, nrows,
                          dd == sd ? hipMemcpyDeviceToDevice : hipMemcpyDefault, dst->ctx->stream));
  HIPCHK(hipStreamSynchronize(dst->ctx->stream));
  return ICS_OK;
}

// Rows of a frame buffer to / from the same buffer of band jobs on OTHER RANKS (one process per GPU, RCCL point-to-point over xGMI):
// the halo exchange and the stop-test gather of lib/banded.py in rank mode.  Whole pitch rows travel (the apron columns are
// zero on both sides).  ics_group_sendrecv_device: ics_group.hip.
int ics_group_sendrecv_device(ics_group* g, const float* send, size_t send_count, int send_peer, float* recv, size_t recv_count, int recv_peer);
extern "C" int ics_rl_exchange_rows(ics_rl* j, ics_group* g, int which, int send_row0, int send_rows, int send_peer, int recv_row0, int recv_rows, int recv_peer) {
  if (!j || !g) return fail(ICS_EINVAL, "NULL argument");
  float* frame; int rows, cols, oy, ox;
  if (frame_of(j, which, &frame, &rows, &cols, &oy, &ox) != 0) return fail(ICS_EINVAL, "buffer %d is not a frame", which);
  if (send_peer >= 0 && (send_row0 < 0 || send_rows < 1 || send_row0 + send_rows > rows)) return fail(ICS_EINVAL, "send rows [%d, %d) of %d", send_row0, send_row0 + send_rows, rows);
  if (recv_peer >= 0 && (recv_row0 < 0 || recv_rows < 1 || recv_row0 + recv_rows > rows)) return fail(ICS_EINVAL, "receive rows [%d, %d) of %d", recv_row0, recv_row0 + recv_rows, rows);
  HIPCHK(hipSetDevice(j->ctx->device));
  HIPCHK(hipStreamSynchronize(j->ctx->stream));          // what is sent has been produced
  if (recv_peer >= 0 && which == ICS_BUF_IMAGE) image_changed(j);
  const size_t pitch = (size_t)j->g.pitch;
  // row r of the buffer = frame row oy + r; a pitch row starts ax pixels left of the frame origin
  float* base = frame;                                  // allocation start of the frame (origin = base + ay rows + ax pixels)
  const float* sp = base + (size_t)(j->g.ay + oy + (send_peer >= 0 ? send_row0 : 0)) * pitch;
  float* rp = base + (size_t)(j->g.ay + oy + (recv_peer >= 0 ? recv_row0 : 0)) * pitch;
  return ics_group_sendrecv_device(g, sp, send_peer >= 0 ? (size_t)send_rows * pitch : 0, send_peer, rp, recv_peer >= 0 ? (size_t)recv_rows * pitch : 0, recv_peer);
}

// -------------------------------------------------------------------------------------------------
// stop-test scratch: Gaussian window weights (pyx:393-404), twiddles, P x P x 3 complex buffer
static int ensure_window(ics_rl* j, const ics_rl_params* p) {
  const int H = p->bottom - p->top, W = p->right - p->left;
  // an empty window: the reference slices error[top:bottom, left:right] into an empty array and every statistic is NaN
  // (numpy warns, pyx:600-601,627-638 do not raise); the stop test then never fires.  Decided BEFORE the cache check: jobs are
  // reused across calls (lib/deconvolution.py keeps them), and window W -> empty window -> W must not leave the flag set.
  j->win_empty = (H < 1 || W < 1);
  if (j->win_empty) return ICS_OK;
  if (j->z && j->wt == p->top && j->wb == p->bottom && j->wl == p->left && j->wr == p->right) return ICS_OK;
  if (p->top < 0 || p->left < 0 || p->bottom > j->g.M || p->right > j->g.N)
    return fail(ICS_EINVAL, "stats window [%d:%d, %d:%d] outside the %dx%d image", p->top, p->bottom, p->left, p->right, j->g.M, j->g.N);
  const int need = 2 * (H > W ? H : W) - 1;
  int P = 2, logP = 1;
  while (P < need) { P <<= 1; ++logP; }
  if (P > 8192) return fail(ICS_ENOSUP, "stats window %dx%d needs a %d-point FFT (max 8192, i.e. windows up to 4096 px a side)", H, W, P);
  // The cached key goes first: if an allocation below fails (z alone is 1.6 GB at P = 8192) the job must not keep the old key with
  // freed or half-built buffers -- jobs are reused (lib/deconvolution.py), and the next run with the previous window would pass the
  // cache check and launch the statistics kernels on them.
  auto drop = [&]() {
    j->wt = j->wb = j->wl = j->wr = -1; j->P = 0; j->logP = 0;
    if (j->z) { j->ctx->pool.release(j->z); j->z = nullptr; }
    if (j->tw) { j->ctx->pool.release(j->tw); j->tw = nullptr; }
    if (j->weights) { j->ctx->pool.release(j->weights); j->weights = nullptr; }
  };
  drop();
  int rc;
  const int fail_at = ics_debug().fail_window_alloc.exchange(0, std::memory_order_relaxed);   // test hook: the fail_at-th allocation fails once
  if ((rc = fail_at == 1 ? fail(ICS_ENOMEM, "stats window: allocation failed (test hook)") : dalloc(j->ctx, &j->z, (size_t)3 * P * P, false)) != ICS_OK) { drop(); return rc; }
  if ((rc = fail_at == 2 ? fail(ICS_ENOMEM, "stats window: allocation failed (test hook)") : dalloc(j->ctx, &j->tw, (size_t)P / 2 + 1, false)) != ICS_OK) { drop(); return rc; }
  if ((rc = fail_at == 3 ? fail(ICS_ENOMEM, "stats window: allocation failed (test hook)") : dalloc(j->ctx, &j->weights, (size_t)H * W, false)) != ICS_OK) { drop(); return rc; }
  std::vector<float2> tw(P / 2 + 1);
  for (int k = 0; k < P / 2; ++k) {
    const double ang = -2.0 * M_PI * (double)k / (double)P;
    tw[k] = make_float2((float)cos(ang), (float)sin(ang));
  }
  tw[P / 2] = make_float2(0.f, 0.f);
  // np.linspace(-1., 1., num, dtype=float32) then gaussian_weight(x, 0, 1) in float (pyx:35-36,397-401)
  auto serie = [](int num, std::vector<float>& out) {
    out.resize(num);
    const double step = num > 1 ? 2.0 / (double)(num - 1) : 0.0;
    const float PI = 3.141592653589793f;
    for (int i = 0; i < num; ++i) {
      double y = (double)i * step + (-1.0);
      if (num > 1 && i == num - 1) y = 1.0;
      const float x = (float)y;
      out[i] = expf(-powf(x - 0.f, 2.f) / (2 * powf(1.f, 2.f))) / (1.f * powf(2 * PI, 0.5f));
    }
  };
  std::vector<float> wi, he, w((size_t)H * W);
  serie(H, wi); serie(W, he);
  double sum = 0.0;
  for (int r = 0; r < H; ++r)
    for (int c = 0; c < W; ++c) { w[(size_t)r * W + c] = sqrtf(wi[r] * he[c]); sum += w[(size_t)r * W + c]; }
  const float fs = (float)sum;
  for (auto& v : w) v = v / fs;
  HIPCHK(hipMemcpyAsync(j->tw, tw.data(), tw.size() * sizeof(float2), hipMemcpyHostToDevice, j->ctx->stream));
  HIPCHK(hipMemcpyAsync(j->weights, w.data(), w.size() * sizeof(float), hipMemcpyHostToDevice, j->ctx->stream));
  HIPCHK(hipStreamSynchronize(j->ctx->stream));  // tw / w are stack-owned host vectors
  j->P = P; j->logP = logP; j->wt = p->top; j->wb = p->bottom; j->wl = p->left; j->wr = p->right;
  return ICS_OK;
}

// ---- launch helpers with optional event bracketing -----------------------------------------------
#define RC0(x) do { int rc0_ = (x); if (rc0_ != ICS_OK) return rc0_; } while (0)
struct Prof {
  ics_rl* j; bool on;
  hipStream_t s = nullptr;              // nullptr: the job's stream
  hipStream_t st() const { return s ? s : j->ctx->stream; }
  int grow() {
    if (j->ev_used + 1 > j->ev.size()) {
      for (int i = 0; i < 64; ++i) { hipEvent_t e; HIPCHK(hipEventCreate(&e)); j->ev.push_back(e); }
    }
    return ICS_OK;
  }
  // Consecutive brackets on one stream share an event: the end of one is the begin of the next (an event record between two dependent
  // kernels is a bubble of a few microseconds on the device; 10 per bracketed blind iteration were 2 % of bench.py's timed region).
  // Anything queued outside a bracket breaks the chain: unbracketed launches come through a disabled Prof, other sites call unchain().
  int begin(int cls) {
    if (!on) { j->ev_chain = -1; return ICS_OK; }
    if (j->ev_chain >= 0 && j->ev_chain_stream == st()) j->ev_open = j->ev_chain;
    else {
      RC0(grow());
      HIPCHK(hipEventRecord(j->ev[j->ev_used], st()));
      j->ev_open = (int)j->ev_used++;
    }
    j->ev_open_cls = cls;
    j->ev_chain = -1;
    return ICS_OK;
  }
  int end() {
    if (!on) return ICS_OK;
    RC0(grow());
    HIPCHK(hipEventRecord(j->ev[j->ev_used], st()));
    j->ev_pairs.push_back({j->ev_open, (int)j->ev_used, j->ev_open_cls});
    j->ev_chain = (int)j->ev_used++; j->ev_chain_stream = st();
    j->ev_open = -1;
    return ICS_OK;
  }
  void unchain() { j->ev_chain = -1; }
  // call after a stream synchronisation
  int collect(double* ms, int* launches) {
    if (!on) return ICS_OK;
    size_t done = 0;
    RC0(collect_range(ms, launches, done, j->ev_pairs.size()));
    j->ev_used = 0; j->ev_pairs.clear(); j->ev_chain = -1;
    return ICS_OK;
  }
  // overlapped runs: the pairs [done, upto) are known to be complete; nothing is recycled until the run ends
  int collect_range(double* ms, int* launches, size_t& done, size_t upto) {
    if (!on) return ICS_OK;
    for (size_t i = done; i < upto; ++i) {
      const ics_rl::EvPair& q = j->ev_pairs[i];
      float t = 0.f;
      HIPCHK(hipEventElapsedTime(&t, j->ev[q.b], j->ev[q.e]));
      ms[q.cls] += t; launches[q.cls] += 1;
    }
    done = upto;
    return ICS_OK;
  }
};


// ---- row bands over several ranks: the two per-iteration reductions, in place on the device (lib/banded.py rank mode) -----------
int ics_group_allreduce_device(ics_group* g, void* buf, size_t count, int kind, hipStream_t stream);   // ics_group.hip
int ics_group_info_local(const ics_group* g);
int ics_group_device(const ics_group* g);   // ics_group.hip
extern "C" int ics_rl_allreduce_keys(ics_rl* j, ics_group* g) {
  if (!j || !g) return fail(ICS_EINVAL, "NULL argument");
  if (!ics_group_info_local(g) && ics_group_device(g) != j->ctx->device)   // (the collective runs on the job's stream with the group's communicator)
    return fail(ICS_EINVAL, "the job lives on device %d, the group's communicator on device %d", j->ctx->device, ics_group_device(g));
  if (j->par != 0) return fail(ICS_ESTATE, "ics_rl_allreduce_keys acts on reduction set 0 (stage API); the job is inside an overlapped run");
  HIPCHK(hipSetDevice(j->ctx->device));
  return ics_group_allreduce_device(g, j->red, 6, 0, j->ctx->stream);      // slot 0: [0..2] max|g_k|, [3..5] max u_k
}
extern "C" int ics_rl_allreduce_gradk(ics_rl* j, ics_group* g) {
  if (!j || !g) return fail(ICS_EINVAL, "NULL argument");
  int rank = 0, world = 1;
  RC(ics_group_info(g, &rank, &world));
  if (world == 1 && ics_group_info_local(g)) return ICS_OK;
  if (!ics_group_info_local(g) && ics_group_device(g) != j->ctx->device)
    return fail(ICS_EINVAL, "the job lives on device %d, the group's communicator on device %d", j->ctx->device, ics_group_device(g));
  HIPCHK(hipSetDevice(j->ctx->device));
  const size_t n = (size_t)3 * j->g.K * j->g.K;
  if (!j->gradk64) RC(dalloc(j->ctx, &j->gradk64, n, false));
  HIPCHK(ics_launch_f32_to_f64(j->gradk, j->gradk64, (long)n, j->ctx->stream));
  RC(ics_group_allreduce_device(g, j->gradk64, n, 1, j->ctx->stream));
  HIPCHK(ics_launch_f64_to_f32(j->gradk64, j->gradk, (long)n, j->ctx->stream));
  return ICS_OK;
}


// ICS_CONV_AUTO: matrix-core kernels where they exist and win (ics_conv_mfma_preferred); env ICS_CONV_PATH=vector|matrix
// overrides AUTO
static bool use_matrix_conv(const ics_rl* j, const ics_rl_params* p) {
  if (!j->bt_conv) return false;
  if (p->conv == ICS_CONV_VECTOR) return false;
  if (p->conv == ICS_CONV_MATRIX) return true;
  const int env = ics_debug().conv_path.load(std::memory_order_relaxed);
  // (ICS_CONV_PATH=fft where the tiles do not run -- single stages, tv_mode 1, PSF sizes above 65: what AUTO would take)
  return env == 2 || ((env == 0 || env == 3) && ics_conv_mfma_preferred(j->g.K));
}

// The FFT-tile pipeline (ics_conv_fft.hip; round 5): A1 / A3 / A11 as LDS-resident 128 x 128 overlap-save transforms on planar mirrors,
// fp32 throughout, with the update pass and the matrix-core PSF gradient on the mirrors as well.  Shipped loop only (tv_mode 0, fuse 0),
// PSF sizes 3 ... 65.  Explicit: conv = ICS_CONV_FFT (also through the stage API); under ICS_CONV_AUTO inside ics_rl_run where it measured
// ahead of the matrix-core kernels (fft_preferred); ICS_CONV_PATH=fft|matrix|vector overrides AUTO.
// (PSF sizes 67 ... 97, opened to the tiles late in round 6 -- the stage functions never depended on the size, only the valid part of a tile
//  shrinks: 62 pixels a side at 67, 32 at 97 -- against the matrix cores' tap blocks, profiles/r06_ab_fft_bigk.txt, non-blind / blind:
//  1024^2 67: 0.659 -> 0.160 / 1.270 -> 0.332;  2048^2 67: 1.894 -> 0.333 / 3.621 -> 0.609;  97: 2.455 -> 0.793 / 5.182 -> 1.494;
//  4096^2 67: 6.774 -> 1.059 / 12.78 -> 1.746;  85: 7.822 -> 1.676 / 14.43 -> 2.895;  97: 9.011 -> 2.848 / 18.46 -> 4.989)
#ifndef ICS_FFT_AUTO_MAX_K
#define ICS_FFT_AUTO_MAX_K 85
#endif
#ifndef ICS_FFT_BLK_MIN_PX
#define ICS_FFT_BLK_MIN_PX 500000L
#endif
static bool fft_preferred(const IcsGeom& g, bool blind) {
  // measured on MI355X (NOTES_r05.md): per-pass time of the transform tiles is set by the tile count (128 - K + 1 valid pixels a side),
  // the Toeplitz matrix-core kernels pay K^2.  scripts/ab_fft.py at the end of round 5, ms per inner iteration, matrix cores -> tiles
  // (non-blind / blind):  1024^2 31: 0.118 -> 0.146 / 0.251 -> 0.192;  1448^2 21: 0.162 -> 0.149 / 0.272 -> 0.252;  31: 0.198 -> 0.131 / 0.331 -> 0.248;
  // 2048^2 17: 0.168 -> 0.183 / level;  19: 0.257 -> 0.181 / 0.427 -> 0.324;  4096^2 15: level / 0.848 -> 0.944 (the fused A11 + A13 kernel);
  // 17: 0.616 -> 0.599 / 1.104 -> 0.937;  19: 0.901 -> 0.597 / 1.545 -> 0.936;  6144^2 17: 1.285 -> 1.233 / 2.388 -> 1.836.
  // Round 6, with A11 + A13 fused into one three-transform unit (k_synth_gradk_fft), scripts/ab_fft.py, matrix cores -> tiles (non-blind / blind):
  // 2048^2 15: 0.164 -> 0.185 / 0.237 -> 0.284 (three rounds of units where 2.004 would do);  17: 0.170 -> 0.183 / 0.315 -> 0.286;  1448^2 17: blind 0.195 -> 0.187;
  // 2900^2 15: level / 0.455 -> 0.431;  17: 0.341 -> 0.311 / 0.576 -> 0.433;  4096^2 9: 0.512 -> 0.574 / 0.730 -> 0.764;  13: level / 0.790 -> 0.766;
  // 15: 0.593 -> 0.579 / 0.834 -> 0.785;  6144^2 15: 1.262 -> 1.199 / 1.835 -> 1.680;  8192^2 15: level / 3.256 -> 2.947.
  // ... and with A1 + A3 as one unit per tile pair as well (mode 2, PSF sizes <= 25), profiles/r06_ab_fft.txt (a slow box: the matrix-core blind 4096^2 / 15 line is 0.851 there):
  // 1448^2 15: 0.097 -> 0.114 / 0.154 -> 0.165;  17: level / 0.195 -> 0.156;  2048^2 9: 0.141 -> 0.158 / 0.210 -> 0.230;  13: level / level;  15: 0.169 -> 0.154 / 0.237 -> 0.250;
  // 17: 0.170 -> 0.151 / 0.315 -> 0.248;  2900^2 9: 0.286 -> 0.265 / 0.403 -> 0.363;  15: 0.332 -> 0.296 / 0.457 -> 0.400;  4096^2 5: level / 0.683 -> 0.643;  9: 0.532 -> 0.479 /
  // 0.732 -> 0.662;  15: 0.627 -> 0.526 / 0.851 -> 0.716;  6144^2 9: 1.140 -> 0.999 / 1.622 -> 1.365;  15: 1.256 -> 1.122 / 1.821 -> 1.502.
  const long px = (long)g.uM * g.uN;
  if (g.K > ICS_FFT_AUTO_MAX_K) return ics_conv_fft_blk_supported(g.K) && px >= ICS_FFT_BLK_MIN_PX;      // tap blocks on the tiles
  if (g.K >= 51) return px >= 500000L;
  if (g.K >= 19) return px >= (blind ? 1000000L : 1500000L);
  if (g.K == 17) return px >= (blind ? 2000000L : 4000000L);
  if (g.K == 15) return px >= (blind ? 6000000L : 4000000L);
  if (g.K >= 9) return px >= 6000000L;
  return px >= 12000000L;
}
// The tile kernels address a channel-planar mirror through ONE raw buffer resource with 32-bit byte offsets (ics_conv_fft.hip make_gbuf:
// num_records 2^31 - 1).  A mirror's rows are padded to 64 floats per plane, so for some shapes it is a little LARGER than the HWC frame it
// mirrors: a frame accepted just under the 2 GiB frame limit can have a mirror whose last plane ends beyond 2^31 bytes -- loads there
// would read 0, stores would be dropped, silently (e.g. 18784 x 9256 with a 33 x 33 PSF).  Such shapes never take the tiles.
static bool fft_mirror_fits(const IcsGeom& g) { return ics_planar_floats(g) * sizeof(float) < 0x7FFFFFFFull; }
static bool pam_on_tiles(const ics_rl* j, const ics_rl_params* p, bool in_run) {   // the routing rule of the PAM kinds (tv_mode 2 / 3)
  if (!ics_conv_fft_supported(j->g.K) || p->fuse || !in_run || !fft_mirror_fits(j->g)) return false;        // (single stages of the TV variants run on the HWC kernels)
  if (p->tv_mode != ICS_TV_PAM_ISO && p->tv_mode != ICS_TV_PAM_COLLAB) return false;
  if (p->conv == ICS_CONV_FFT) return true;
  if (p->conv != ICS_CONV_AUTO) return false;
  const int env = ics_debug().conv_path.load(std::memory_order_relaxed);
  if (env == 3) return true;
  // bench.py --tv-mode 2, ICS_CONV_PATH=matrix -> fft (measured with the convolutions and the gradient on the tiles only), ms per inner iteration: 1448^2 / 31 blind 0.357 -> 0.297; 2048^2 / 21 blind 0.470 -> 0.396,
  // non-blind 0.271 -> 0.259; 4096^2 / 19 blind 1.563 -> 1.132; 1100^2 / 45 blind 0.432 -> 0.310
  // round 6: the shipped loop's thresholds -- the same kernels but for the epilogue's operands -- except below 6 Mpx at 15 x 15 and less, where the
  // TV term's own pass on the planes tips it back (bench.py --tv-mode 2, ms per inner iteration, matrix cores -> tiles: 4096^2 / 15 blind 0.873 -> 0.780,
  // tv_mode 3 0.869 -> 0.794; 2048^2 / 15 non-blind 0.176 -> 0.190, blind 0.261 -> 0.293)
  const long px = (long)j->g.uM * j->g.uN;
  return env == 0 && fft_preferred(j->g, p->blind != 0) && (j->g.K >= 17 || px >= 6000000L);
}
static bool use_fft_pipeline(const ics_rl* j, const ics_rl_params* p, bool in_run) {
  if (p->tv_mode != ICS_TV_SHIPPED) return pam_on_tiles(j, p, in_run);   // PAM kinds: TV term, back-projection epilogue and update on the mirrors too
  if (!(ics_conv_fft_supported(j->g.K) || ics_conv_fft_blk_supported(j->g.K)) || p->fuse || !fft_mirror_fits(j->g)) return false;      // (87 ... 255: tap blocks on the tiles)
  if (p->conv == ICS_CONV_FFT) return true;
  if (p->conv != ICS_CONV_AUTO || !in_run) return false;
  const int env = ics_debug().conv_path.load(std::memory_order_relaxed);
  if (env == 3) return true;
  return env == 0 && fft_preferred(j->g, p->blind != 0);
}

// tv_mode 1 rewrites the image in every inner iteration (pyx:547-549 live): a copy would have to be rebuilt each time
static bool use_image_acc(const ics_rl_params* p) {
  return p->tv_mode != ICS_TV_MM_ACTIVE && ics_debug().planar_image.load(std::memory_order_relaxed) != 0;
}

// The run-time-sized fp32 kernels (ics_big.hip) are the only ones above 63, and under ICS_CONV_AUTO they take over from the packed-fp32
// kernels compiled per size wherever the matrix-core kernels are not built or not chosen (51 .. 63; 39 .. 49 with ICS_CONV_PATH=...): they
// beat them (shipped loop, 2048^2 non-blind, ms per pass, compiled -> run-time-sized): synthesis
// 39: 0.626 -> 0.594, 45: 0.829 -> 0.756, 55: 1.238 -> 1.013, 63: 1.71 -> 1.27; back-projection (incl. the separate maxima pass)
// 39: 0.975 -> 0.716, 45: 1.348 -> 0.915, 55: 2.109 -> 1.233, 63: 2.88 -> 1.52.  ICS_CONV_VECTOR keeps the compiled kernels
// (tests/test_gpu_stages.py drives them at every size).
static bool use_big_conv(const ics_rl* j, const ics_rl_params* p, int mode) {
  const int K = j->g.K;
  if (ics_big_supported(K)) return true;
  if (mode == 2 || p->conv != ICS_CONV_AUTO || p->tv_mode != ICS_TV_SHIPPED || K < 39) return false;
  return ics_debug().conv_path.load(std::memory_order_relaxed) != 1;   // ICS_CONV_PATH=vector: as ICS_CONV_VECTOR
}

// PSF sizes 51 ... 127 on the matrix cores (ICS_CONV_AUTO / ICS_CONV_MATRIX, shipped loop): a convolution is linear in its taps, so
// the K x K PSF is cut into blk_n x blk_n blocks of Kb x Kb taps (Kb odd, <= 33: sizes k_conv_mfma is built for); block (qa, qb) is
//     out_q[y][x] = sum_{a',b' < Kb} W[qa Kb + a'][qb Kb + b'] * in'[y + a' - Kb/2][x + b' - Kb/2],   in' = in shifted by (qa Kb + Kb/2 - pad, ...)
// i.e. the Kb x Kb kernel on a shifted pointer with a geometry that differs in K and pad only (mode 0 tiles start at the kernel's
// own pad: all three pointers move by pad - Kb/2 so that this is the image origin).  The blocks are summed as a chain through the
// kernels' own operand frames, alternating between the result frame and a scratch frame (do_conv_blocks); the maxima of A7 are taken
// over the sum by k_band_reduce.  2048^2, 63 x 63: 1.24 / 1.50 ms (run-time-sized fp32 kernel) -> see NOTES_r03.md 4c.
static bool use_block_conv(const ics_rl* j, const ics_rl_params* p, int mode) {
  if (!j->blk_conv || mode == 2 || p->tv_mode != ICS_TV_SHIPPED || p->conv == ICS_CONV_VECTOR) return false;
  if (psf_blocks_only(j->g.K)) return true;   // (no other path: ICS_CONV_PATH does not apply)
  return p->conv == ICS_CONV_MATRIX || ics_debug().conv_path.load(std::memory_order_relaxed) != 1;
}

static int do_conv_blocks(ics_rl* j, int mode, const ics_rl_params* p, int slot, Prof& pr) {
  const IcsGeom& G = j->g;
  const int Kb = j->blk_kb, nb = j->blk_n, padb = Kb / 2, pad = G.pad;
  const size_t tf = ics_conv_mfma_table_floats(Kb);
  float* out = org(j, mode == 1 ? j->gr : j->e);
  const float* in = org(j, mode == 1 ? j->e : j->u);
  const ptrdiff_t oshift = mode == 0 ? (ptrdiff_t)(pad - padb) * (G.pitch + 3) : 0;   // mode 0: the kernel's tile origin (padb, padb) = the image origin
  // Mode 0 (residual = synthesis - image): the blocks form a CHAIN through the kernel's own "- image" operand instead of being added by
  // passes of their own.  With S_q = C_0 + ... + C_q - image, block q stores T_q = s_q S_q = conv(s_q W_q) - T_{q-1}, signs alternating
  // (s_q = -s_{q-1}: the weight tables of the conv orientation carry them, k_pack_blocks) and ending on s_{n-1} = +1; the chain starts from
  // T_{-1} = s_0 * image: the image itself for an odd number of blocks, a negated copy of it (blk_negf) for an even one.  T_q alternates
  // between the result frame and the scratch frame so that the last one lands in the result.  Same sums in the same order as adding the
  // blocks one by one (IEEE negation is exact and rounding is symmetric), one launch and 36 B/px less per block after the first, and no
  // frame of zeros read as the image operand.  (255 x 255 at 2048^2: synthesis 8.8 -> see NOTES_r04.md 4d.)
  const int nq = nb * nb;
  if (mode == 0 && !(nq & 1) && !j->negf_valid) {
    HIPCHK(ics_launch_frame_neg(j->blk_negf, j->f, j->frame_floats, j->ctx->stream));
    j->negf_valid = true;
  }
  RC(pr.begin(mode == 0 ? ICS_K_SYNTH : ICS_K_BACKPROJECT));
  for (int q = 0; q < nq; ++q) {
    const int qa = q / nb, qb = q - qa * nb;
    IcsConvArgs a;
    a.g = G; a.g.K = Kb; a.g.pad = padb;
    a.lambd = p->lambd; a.w = nullptr; a.tv = nullptr; a.tv_kind = 0;
    a.in = in + oshift + (ptrdiff_t)(qa * Kb + padb - pad) * G.pitch + 3 * (qb * Kb + padb - pad);
    if (mode == 0) {
      float* const fr[2] = {out, org(j, j->blk_scr)};            // T_q -> fr[(nq - 1 - q) & 1], read from the other one
      a.out = fr[(nq - 1 - q) & 1] + oshift;
      a.f = (q == 0 ? ((nq & 1) ? org(j, j->f) : org(j, j->blk_negf)) : fr[(nq - q) & 1]) + oshift;
    } else {
      // Mode 1 chains through the epilogue the PAM kinds use: with tv_kind = 2 the kernel stores fl32(double(T) + double(lambd * sums)) for
      // an operand frame T -- lambd = 1 and T = the blocks so far make that T_q = T_{q-1} + C_q, the sum the adding pass formed.  The
      // first block stores its raw sums as before; the maxima of A7 are taken over the finished frame by k_band_reduce below.
      float* const fr[2] = {out, org(j, j->blk_scr)};
      a.out = fr[(nq - 1 - q) & 1];
      a.f = org(j, j->f);                                       // (mode 1 reads no image operand)
      if (q > 0) { a.tv = fr[(nq - q) & 1]; a.tv_kind = ICS_TV_PAM_ISO; a.lambd = 1.0f; }
    }
    a.u = org(j, j->u); a.ut = org(j, ut_of(j));
    a.red = j->blk_red;                                   // (per-block maxima mean nothing)
    a.gr = nullptr; a.u_out = nullptr; a.scal = j->scal; a.dofkeys = dof_of(j);
    a.step = p->step_factor; a.blind = p->blind; a.want_dof = 0;
    a.bt = (mode == 1 ? j->blk_corr : j->blk_conv) + (size_t)q * tf;
    a.facc[0] = a.facc[1] = nullptr;
    a.sched = j->sched;
    HIPCHK(ics_launch_conv_mfma(mode, a, j->ctx->stream));
  }
  if (mode == 1) HIPCHK(ics_launch_band_reduce(out, org(j, j->u), org(j, ut_of(j)), G, p->lambd, 0, G.uM, red_of(j) + slot * ICS_RED_STRIDE, j->ctx->stream));
  RC(pr.end());
  return ICS_OK;
}

static int do_conv_fft(ics_rl* j, int mode, const ics_rl_params* p, int slot, Prof& pr) {
  IcsConvArgs a;
  memset(&a, 0, sizeof a);
  a.g = j->g; a.lambd = p->lambd;
  if (mode == 1) { a.in = porg(j, j->e); a.out = porg(j, j->gr); }
  else { a.in = porg(j, j->u); a.out = porg(j, j->e); }
  a.f = porg(j, j->f); a.u = porg(j, j->u); a.ut = porg(j, ut_of(j));
  a.red = red_of(j) + slot * ICS_RED_STRIDE;
  a.step = p->step_factor; a.blind = p->blind;
  if (p->tv_mode >= ICS_TV_PAM_ISO && mode == 1) {   // PAM kinds: the epilogue takes u and T, stores G = T + lambd gradu
    a.tv = j->tvf ? porg(j, j->tvf) : nullptr; a.tv_kind = p->tv_mode;
    if (!a.tv) return fail(ICS_ESTATE, "FFT pipeline: the TV frame has no planar mirror");
  }
  if (!a.in || !a.out || !a.f || !a.u || !a.ut) return fail(ICS_ESTATE, "FFT pipeline: a frame has no planar mirror");
  RC(pr.begin(mode == 0 ? ICS_K_SYNTH : ICS_K_BACKPROJECT));
  if (ics_conv_fft_blk_supported(j->g.K)) {   // 87 ... 255: tap blocks, their products summed in the frequency domain (k_conv_fft_blk)
    int nb, kb;
    ics_conv_fft_blk_shape(j->g.K, &nb, &kb);
    HIPCHK(ics_launch_conv_fft_blk(mode, a, mode == 1 ? j->spec_corr : j->spec_conv, nb, kb, j->ctx->stream));
  } else
  HIPCHK(ics_launch_conv_fft(mode, a, mode == 1 ? j->spec_corr : j->spec_conv, ICS_FFT_PL_ALL, j->ctx->stream));
  RC(pr.end());
  return ICS_OK;
}

// Mode 2 of the tiles: A1 + A2 + A3 in ONE unit per tile pair (k_conv_fft<2>): interior tiles stay in the frequency domain between the two
// convolutions -- one forward and one inverse transform instead of two of each, at 128 - 2 K + 2 valid pixels a side instead of 128 - K + 1,
// which pays for small PSFs.  Inside ics_rl_run only (the residual frame is not produced: the statistics' window of it comes from a
// window-sized launch of mode 0 where the loop does not rewrite it anyway), shipped loop only.
// scripts/ab_conv2.py on MI355X, ms per inner iteration, two kernels -> one unit (non-blind / blind): 1024^2 9: 0.093 -> 0.080 / 0.144 -> 0.125;  15: 0.101 -> 0.085 /
// 0.156 -> 0.136;  25: 0.109 -> 0.100 / 0.172 -> 0.161;  2048^2 15: 0.186 -> 0.154 / 0.293 -> 0.250;  21: 0.187 -> 0.178 / 0.294 -> 0.281;  25: 0.201 -> 0.175 / 0.319 -> 0.283;
// 31: 0.184 -> 0.230 / 0.306 -> 0.351;  4096^2 9: 0.572 -> 0.470 / 0.775 -> 0.668;  15: 0.581 -> 0.518 / 0.797 -> 0.714 (another box: 0.576 -> 0.485 / 0.794 -> 0.686);
// 21: 0.600 -> 0.561 / 0.828 -> 0.774;  25: level / 0.867 -> 0.859;  31: 0.636 -> 0.735 / 0.909 -> 0.990.
#ifndef ICS_CONV2_MAX_K
#define ICS_CONV2_MAX_K 25
#endif
static bool use_conv2(const ics_rl* j, const ics_rl_params* p) {
  if (!j->fft_on || p->tv_mode == ICS_TV_MM_ACTIVE || p->fuse) return false;      // (shipped loop and the PAM kinds, whose epilogue takes u and T)
  const int sw = ics_debug().fft_conv2.load(std::memory_order_relaxed);
  if (sw == 0 || j->conv2_off || !ics_conv2_fft_supported(j->g)) return false;
  return sw == 2 || j->g.K <= ICS_CONV2_MAX_K;
}
static int do_conv2(ics_rl* j, const ics_rl_params* p, int slot, Prof& pr) {
  if (!j->fspec) RC(dalloc(j->ctx, &j->fspec, ics_conv2_fft_fspec_floats(j->g), false));
  if (!j->fspec_valid) {
    if (!porg(j, j->f)) return fail(ICS_ESTATE, "FFT pipeline: the image has no planar mirror");
    HIPCHK(ics_launch_fft_image_spectrum(porg(j, j->f), j->g, j->fspec, j->ctx->stream));
    j->fspec_valid = true;
  }
  IcsConvArgs a;
  memset(&a, 0, sizeof a);
  a.g = j->g; a.lambd = p->lambd;
  a.in = porg(j, j->u); a.out = porg(j, j->gr); a.f = porg(j, j->f); a.u = porg(j, j->u); a.ut = porg(j, ut_of(j));
  a.red = red_of(j) + slot * ICS_RED_STRIDE;
  a.step = p->step_factor; a.blind = p->blind;
  if (p->tv_mode >= ICS_TV_PAM_ISO) {   // PAM kinds: the epilogue takes u and T, stores G = T + lambd gradu
    a.tv = j->tvf ? porg(j, j->tvf) : nullptr; a.tv_kind = p->tv_mode;
    if (!a.tv) return fail(ICS_ESTATE, "FFT pipeline: the TV frame has no planar mirror");
  }
  if (!a.in || !a.out || !a.f || !a.ut) return fail(ICS_ESTATE, "FFT pipeline: a frame has no planar mirror");
  RC(pr.begin(ICS_K_SYNTH_BACKPROJECT));
  HIPCHK(ics_launch_conv2_fft(a, j->spec_conv, j->spec_corr, j->fspec, j->ctx->stream));
  RC(pr.end());
  return ICS_OK;
}
// A1 + A2 over the stop-test window alone (the tiles that cover image rows [top, bottom) x columns [left, right)): what A18 / A19 read of the
// residual (pyx:600-601, 627) when mode 2 ran the iteration
static int do_conv_fft_window(ics_rl* j, const ics_rl_params* p, Prof& pr) {
  IcsConvArgs a;
  memset(&a, 0, sizeof a);
  a.g = j->g; a.lambd = p->lambd;
  a.in = porg(j, j->u); a.out = porg(j, j->e); a.f = porg(j, j->f); a.u = a.in; a.ut = a.in;
  if (!a.in || !a.out || !a.f) return fail(ICS_ESTATE, "FFT pipeline: a frame has no planar mirror");
  const int pad = j->g.pad;
  int y0 = p->top + pad, y1 = p->bottom + pad, x0 = p->left + pad, x1 = p->right + pad;
  if (y0 < pad) y0 = pad; if (x0 < pad) x0 = pad; if (y1 > pad + j->g.M) y1 = pad + j->g.M; if (x1 > pad + j->g.N) x1 = pad + j->g.N;
  if (y0 >= y1 || x0 >= x1) return ICS_OK;
  RC(pr.begin(ICS_K_SYNTH));
  HIPCHK(ics_launch_conv_fft_region(a, j->spec_conv, y0, x0, y1, x1, j->ctx->stream));
  RC(pr.end());
  return ICS_OK;
}

static int do_conv(ics_rl* j, int mode, const ics_rl_params* p, int slot, int want_dof, Prof& pr) {
  if (j->fft_on && mode != 2) return do_conv_fft(j, mode, p, slot, pr);
  if (use_block_conv(j, p, mode)) return do_conv_blocks(j, mode, p, slot, pr);
  IcsConvArgs a;
  a.g = j->g; a.lambd = p->lambd;
  if (mode == 1) { a.in = org(j, j->e); a.w = j->wcorr; a.out = org(j, j->gr); }
  else { a.in = org(j, j->u); a.w = j->wconv; a.out = org(j, j->e); }
  a.f = org(j, j->f); a.u = org(j, j->u); a.ut = org(j, ut_of(j));
  a.red = red_of(j) + slot * ICS_RED_STRIDE;
  a.gr = org(j, j->gr); a.u_out = org(j, j->u2); a.scal = j->scal; a.dofkeys = dof_of(j);
  a.tv = (p->tv_mode != ICS_TV_SHIPPED && j->tvf) ? org(j, j->tvf) : nullptr; a.tv_kind = a.tv ? p->tv_mode : 0;
  a.step = p->step_factor; a.blind = p->blind; a.want_dof = want_dof;
  const bool matrix = mode != 2 && use_matrix_conv(j, p);
  a.bt = matrix ? (mode == 1 ? j->bt_corr : j->bt_conv) : nullptr;
  a.facc[0] = a.facc[1] = nullptr;
  if (matrix && mode == 0 && use_image_acc(p)) {   // the image operand of the residual in accumulator order (ics_image_acc.h)
    const int rs = ics_conv_mfma_rs(j->g.K, j->g);
    if (rs) { RC(ensure_image_acc(j, rs)); a.facc[rs == 2 ? 0 : 1] = j->facc[rs == 2 ? 0 : 1]; }
  }
  a.sched = matrix ? j->sched : nullptr;   // counters of the dynamic tile walk: the launcher decides per launch (ics_conv_mfma.hip)
  RC(pr.begin(mode == 0 ? ICS_K_SYNTH : (mode == 1 ? ICS_K_BACKPROJECT : ICS_K_UPDATE_SYNTH)));
  if (!matrix && use_big_conv(j, p, mode)) {   // run-time-sized kernels (ics_big.hip); the maxima of A7 as a pass of their own
    HIPCHK(ics_launch_conv_big(mode, a, j->psf, j->ctx->stream));
    if (mode == 1) HIPCHK(ics_launch_band_reduce(a.out, a.u, a.ut, j->g, p->lambd, 0, j->g.uM, a.red, j->ctx->stream));
  } else if (matrix) HIPCHK(ics_launch_conv_mfma(mode, a, j->ctx->stream));
  else HIPCHK(ics_launch_conv(mode, a, j->ctx->stream));
  RC(pr.end());
  if (mode == 2) { float* t = j->u; j->u = j->u2; j->u2 = t; }  // the updated frame is now `u`
  return ICS_OK;
}

static int do_update(ics_rl* j, const ics_rl_params* p, int slot, int want_dof, Prof& pr) {
  IcsUpdateArgs a;
  a.u = org(j, j->u); a.ut = org(j, ut_of(j)); a.g = org(j, j->gr); a.f = org(j, j->f);
  a.u_out = org(j, j->ut_is_u ? j->u2 : j->u);
  a.red = red_of(j) + slot * ICS_RED_STRIDE; a.scal = j->scal; a.dofkeys = dof_of(j);
  a.tv = (p->tv_mode != ICS_TV_SHIPPED && j->tvf) ? org(j, j->tvf) : nullptr; a.tv_kind = a.tv ? p->tv_mode : 0; a.f_rw = org(j, j->f);
  a.step = p->step_factor; a.lambd = p->lambd; a.blind = p->blind; a.want_dof = want_dof; a.geo = j->g;
  if (a.tv_kind == ICS_TV_MM_ACTIVE) image_changed(j);          // pyx:547-549: this update also steps the image
  RC(pr.begin(ICS_K_UPDATE));
  if (j->fft_on) {   // the same pass on the planar mirrors (bit-identical arithmetic, ics_planar.hip)
    a.u = porg(j, j->u); a.ut = porg(j, ut_of(j)); a.g = porg(j, j->gr); a.f = porg(j, j->f);
    a.u_out = porg(j, j->ut_is_u ? j->u2 : j->u); a.f_rw = nullptr;
    if (!a.u || !a.ut || !a.g || !a.f || !a.u_out) return fail(ICS_ESTATE, "FFT pipeline: a frame has no planar mirror");
    HIPCHK(ics_launch_update_planar(a, j->ctx->stream));
  } else
  HIPCHK(ics_launch_update(a, j->ctx->stream));
  RC(pr.end());
  if (j->ut_is_u) {  // rotate: the untouched old u is the majoriser now, the stale ut frame becomes the spare
    float* old_ut = j->ut;
    j->ut = j->u; j->u = j->u2; j->u2 = old_ut;
    j->ut_is_u = false;
  }
  return ICS_OK;
}

static int ensure_tv(ics_rl* j) {
  if (j->tvf) return ICS_OK;
  int rc = dalloc(j->ctx, &j->tvf, j->frame_floats);
  return rc;
}

static int do_tvterm(ics_rl* j, const ics_rl_params* p, int slot, Prof& pr) {
  IcsTvTermArgs a;
  a.u = org(j, j->u); a.ut = org(j, ut_of(j)); a.f = org(j, j->f); a.tv = org(j, j->tvf);
  a.red = red_of(j) + slot * ICS_RED_STRIDE; a.epsilon = p->blind ? 1e-2f : 1e-6f; a.kind = p->tv_mode; a.geo = j->g;
  if (j->fft_on) {   // PAM kinds on the planar mirrors (the kernel reads u and writes T; ut and f are not touched by these kinds)
    a.u = porg(j, j->u); a.tv = porg(j, j->tvf); a.planar = 1;
    if (!a.u || !a.tv) return fail(ICS_ESTATE, "FFT pipeline: a frame has no planar mirror");
  }
  RC(pr.begin(ICS_K_UPDATE));   // accounted with the elementwise class
  HIPCHK(ics_launch_tvterm(a, j->ctx->stream));
  RC(pr.end());
  return ICS_OK;
}

// The PSF gradient has its own choice: the matrix-core kernel wins at every size it is built for (K <= 31; at K = 19, 21,
// where AUTO keeps the packed-fp32 convolutions, 0.35 vs 0.97 ms at 4096^2)
static bool use_matrix_gradk(const ics_rl* j, const ics_rl_params* p) {
  if (!ics_gradk_mfma_supported(j->g.K) || p->conv == ICS_CONV_VECTOR) return false;
  if (p->conv == ICS_CONV_MATRIX || j->fft_on) return true;   // (the FFT pipeline's gradient reads the mirrors: matrix-core kernel only)
  return ics_debug().conv_path.load(std::memory_order_relaxed) != 1;
}

// PSF sizes 33 ... 127: the fp16-split matrix-core gradient is built for up to 31 x 31 taps (2 x 2 blocks of 16; a third block row does
// not fit its registers), the fp32-MFMA kernel that took over above runs at a third of its rate.  The gradient is a sum over pixels
// per tap, so the K x K taps split into blocks of at most 31 x 31 -- rows / columns [0, 31), [31, 62), ... -- each a gradient of its own:
//   gradk[a0 + a'][b0 + b'] = sum E[y][x] * U'[y + pad' - a'][x + pad' - b'],   U' = U shifted by (pad - a0 - pad', pad - b0 - pad')
// i.e. the 31 x 31 (or, for the small corner block, 15 x 15) kernel on a shifted frame pointer with a geometry that differs in K and
// pad only; the frames' aprons (16 * ceil(K / 16) rows / pixels) cover the shifts.  4096^2, 45 x 45: 1.92 -> 1.22 ms.
static bool use_split_gradk(const ics_rl* j, const ics_rl_params* p) {
  const int K = j->g.K;
  if (K < 33 || p->conv == ICS_CONV_VECTOR) return false;
  if (psf_blocks_only(K) || j->fft_on) return true;
  return p->conv == ICS_CONV_MATRIX || ics_debug().conv_path.load(std::memory_order_relaxed) != 1;
}

// (33 ... 49: 2 x 2 blocks; 51 ... 127: up to 5 x 5 -- 2048^2: 63 x 63 0.98 -> 0.8 ms, 65 x 65 3.3 -> 0.8, 127 x 127 16.3 -> 2.3)
static int do_gradk_split(ics_rl* j, Prof& pr) {
  const int K = j->g.K, pad = j->g.pad, L = 31, n = (K + L - 1) / L;
  RC(pr.begin(ICS_K_PSF_GRADIENT));
  for (int qa = 0; qa < n; ++qa)
    for (int qb = 0; qb < n; ++qb) {
      const int a0 = qa * L, b0 = qb * L, La = K - a0 < L ? K - a0 : L, Lb = K - b0 < L ? K - b0 : L;
      const int Ks = (La <= 15 && Lb <= 15) ? 15 : 31, pads = Ks / 2, nt = Ks == 15 ? 16 : 32;
      IcsGradkArgs a;
      a.geo = j->g; a.geo.K = Ks; a.geo.pad = pads;
      a.planar = j->fft_on ? 1 : 0;
      if (a.planar) { a.e = porg(j, j->e); a.u = porg(j, j->u) + (ptrdiff_t)(pad - a0 - pads) * ics_ppitch(j->g) + (pad - b0 - pads); }
      else { a.e = org(j, j->e); a.u = org(j, j->u) + (ptrdiff_t)(pad - a0 - pads) * j->g.pitch + 3 * (pad - b0 - pads); }
      a.partial = j->partial;
      // as many persistent workgroups per CU as for the sizes the kernel was built for, within what the partial buffer (sized for K) holds
      int nblocks = 2 * j->ctx->cus;                        // (GCfg::WGS of ics_gradk_mfma.hip)
      if (const int mw = ics_debug().max_wgs.load(std::memory_order_relaxed); mw > 0 && nblocks > mw) nblocks = mw;
      const long cap = (long)(j->partial_floats / (3UL * nt * nt));
      if (nblocks > cap) nblocks = (int)cap;
      HIPCHK(ics_launch_gradk_mfma(a, nblocks, j->ctx->stream));
      HIPCHK(ics_launch_gradk_reduce_block(j->partial, nblocks, j->gradk, nt, La, Lb, K, a0, b0, j->ctx->stream));
    }
  RC(pr.end());
  return ICS_OK;
}

// the FFT-tile pipeline's own PSF gradient (k_gradk_fft): two forward transforms per tile pair, products added up in the frequency domain
static bool use_fft_gradk(const ics_rl* j) { return j->fft_on && ics_debug().fft_gradk.load(std::memory_order_relaxed) != 0; }

static int do_gradk(ics_rl* j, const ics_rl_params* p, Prof& pr) {
  if (use_fft_gradk(j)) {
    RC(pr.begin(ICS_K_PSF_GRADIENT));
    if (ics_conv_fft_blk_supported(j->g.K)) {   // one launch per block of lags
      int nb, kb;
      ics_conv_fft_blk_shape(j->g.K, &nb, &kb);
      if ((size_t)ics_gradk_fft_blocks(j->ctx->cus) * kb * kb > j->partial_floats) return fail(ICS_ESTATE, "PSF gradient scratch too small for the lag blocks");
      HIPCHK(ics_launch_gradk_fft_blk(porg(j, j->u), porg(j, j->e), j->g, nb, kb, j->partial, j->gradk, j->ctx->stream));
    } else
    HIPCHK(ics_launch_gradk_fft(porg(j, j->u), porg(j, j->e), j->g, j->partial, j->gradk, j->ctx->stream));
    RC(pr.end());
    return ICS_OK;
  }
  if (use_split_gradk(j, p)) return do_gradk_split(j, pr);
  IcsGradkArgs a;
  a.e = org(j, j->e); a.u = org(j, j->u); a.partial = j->partial; a.geo = j->g; a.planar = 0;
  if (j->fft_on) { a.e = porg(j, j->e); a.u = porg(j, j->u); a.planar = 1; }
  RC(pr.begin(ICS_K_PSF_GRADIENT));
  if (ics_big_supported(j->g.K) && !j->fft_on) HIPCHK(ics_launch_gradk_big(a, j->gradk_blocks, j->ctx->stream));
  else if (use_matrix_gradk(j, p)) HIPCHK(ics_launch_gradk_mfma(a, j->gradk_blocks, j->ctx->stream));
  else HIPCHK(ics_launch_gradk(a, j->gradk_blocks, j->ctx->stream));
  HIPCHK(ics_launch_gradk_reduce(j->partial, j->gradk_blocks, j->gradk, j->g, j->ctx->stream));
  RC(pr.end());
  return ICS_OK;
}

// A11 + A13 as one three-transform unit on the tiles (k_synth_gradk_fft): wherever the pipeline takes its gradient on the tiles
static bool use_fused_fft(const ics_rl* j, const ics_rl_params* p) {
  return use_fft_gradk(j) && ics_conv_fft_supported(j->g.K) && !p->fuse && !(p->flags & ICS_FLAG_NO_FUSED_GRADK) && ics_debug().fft_fused.load(std::memory_order_relaxed) != 0;
}
static int do_synth_gradk_fft(ics_rl* j, const ics_rl_params* p, int store_all, Prof& pr) {
  const float *u = porg(j, j->u), *f = porg(j, j->f);
  float* e = porg(j, j->e);
  if (!u || !f || !e) return fail(ICS_ESTATE, "FFT pipeline: a frame has no planar mirror");
  const int pad = j->g.pad;
  RC(pr.begin(ICS_K_SYNTH_GRADK));
  HIPCHK(ics_launch_synth_gradk_fft(u, f, e, j->spec_conv, j->g, p->top + pad, p->bottom + pad, p->left + pad, p->right + pad, store_all, j->partial, j->gradk, j->ctx->stream));
  RC(pr.end());
  return ICS_OK;
}

#ifndef ICS_FUSED_DEFAULT_RS
#define ICS_FUSED_DEFAULT_RS 4
#endif
// A11 + A13 in one kernel where it exists (matrix-core path, MK <= 15): ics_synth_gradk_mfma.hip
static bool use_fused_gradk(const ics_rl* j, const ics_rl_params* p) {
  if (!ics_synth_gradk_supported(j->g.K) || !j->bt_conv || j->fft_on) return false;
  if (p->flags & ICS_FLAG_NO_FUSED_GRADK) return false;
  return ics_debug().fused_gradk.load(std::memory_order_relaxed) != 0 && use_matrix_conv(j, p) && use_matrix_gradk(j, p);
}

static int do_synth_gradk(ics_rl* j, const ics_rl_params* p, int store_all, Prof& pr) {
  IcsFusedArgs a;
  a.u = org(j, j->u); a.f = org(j, j->f); a.e_out = org(j, j->e); a.bt = j->bt_conv; a.partial = j->partial; a.g = j->g;
  a.wy0 = p->top + j->g.pad; a.wy1 = p->bottom + j->g.pad; a.wx0 = p->left + j->g.pad; a.wx1 = p->right + j->g.pad;
  a.store_all = store_all;
  // tile height: 32-row tiles with three workgroups per CU, or 64-row tiles with two (ics_synth_gradk_mfma.hip); debug switch fused_rs
  // Measured on MI355X (blind, ms per inner iteration, 64-row -> 32-row form): 255^2 (deblur_module's blind window: 16 tiles of 64 x 64 on
  // 256 CUs) 0.1095 -> 0.0985, 1024^2 0.1415 -> 0.1316, 2048^2 0.275 -> 0.266; 4096^2 level (NOTES_r03.md 4c).  Hence 32-row tiles up to
  // 2500 tiles of 64 x 64 (~3200^2), 64-row tiles above.
  const int frs = ics_debug().fused_rs.load(std::memory_order_relaxed);
  a.rs = frs == 2 || frs == 4 ? frs : ((long)j->g.tiles_x * j->g.tiles_y <= 2500 ? 2 : ICS_FUSED_DEFAULT_RS);
  a.facc = nullptr;
  if (use_image_acc(p)) { RC(ensure_image_acc(j, a.rs)); a.facc = j->facc[a.rs == 2 ? 0 : 1]; }
  int nblocks = j->gradk_blocks;
  if (a.rs == 2) {
    const int tiles32 = ((j->g.N + 63) / 64) * ((j->g.M + 31) / 32);
    nblocks = j->fused2_blocks < tiles32 ? j->fused2_blocks : tiles32;
  }
  RC(pr.begin(ICS_K_SYNTH_GRADK));
  HIPCHK(ics_launch_synth_gradk(a, nblocks, j->ctx->stream));
  HIPCHK(ics_launch_gradk_reduce(j->partial, nblocks, j->gradk, j->g, j->ctx->stream));
  RC(pr.end());
  return ICS_OK;
}

static int do_psf(ics_rl* j, const ics_rl_params* p, Prof& pr) {
  RC(pr.begin(ICS_K_PSF_UPDATE));
  RC(pack_weights(j, 1, p->step_factor, p->correlation, j->ctx->stream));
  RC(pr.end());
  return ICS_OK;
}

static int do_majorize(ics_rl* j, Prof& pr) {
  j->ut_is_u = false;
  RC(pr.begin(ICS_K_MAJORIZE));
  HIPCHK(hipMemcpyAsync(j->ut, j->u, j->frame_floats * 4, hipMemcpyDeviceToDevice, j->ctx->stream));
  RC(pr.end());
  return ICS_OK;
}

static int do_stats(ics_rl* j, const ics_rl_params* p, Prof& pr, int rearm = 0, hipStream_t st = nullptr) {
  if (!st) st = j->ctx->stream;
  if (j->win_empty) {
    static const float nan3[3] = {nanf(""), nanf(""), nanf("")};   // M_r, Hu, varu
    HIPCHK(hipMemcpyAsync(j->scal + ICS_SC_MR, nan3, sizeof nan3, hipMemcpyHostToDevice, j->ctx->stream));
    return ICS_OK;
  }
  if (j->fft_on) {   // A18 / A19 read HWC frames: bring the window of e and u over from the mirrors (u-frame rows [top, bottom + 2 pad))
    const int pad2 = 2 * j->g.pad;
    HIPCHK(ics_launch_planar_convert(false, pl_of(j, j->e), j->e, j->g, false, p->top, p->bottom + pad2, p->left, p->right + pad2, st));
    HIPCHK(ics_launch_planar_convert(false, pl_of(j, j->u), j->u, j->g, false, p->top, p->bottom + pad2, p->left, p->right + pad2, st));
  }
  IcsStatsArgs a;
  a.e = org(j, j->e); a.u = org(j, j->u); a.scal = j->scal; a.dofkeys = dof_of(j); a.dacc = j->dacc; a.ukey = j->ukey;
  a.z = j->z; a.tw = j->tw; a.weights = j->weights;
  a.top = p->top; a.bottom = p->bottom; a.left = p->left; a.right = p->right;
  a.P = j->P; a.logP = j->logP; a.do_mr = p->stop_test != 0; a.geo = j->g;
  a.red = red_of(j); a.rearm = rearm;
  RC(pr.begin(ICS_K_STATS));
  HIPCHK(ics_launch_stats(a, st));
  RC(pr.end());
  return ICS_OK;
}

static int reset_dofkeys(ics_rl* j) {
  static const uint32_t init[4] = {0xFFFFFFFFu, 0u, 0u, 0u};
  HIPCHK(hipMemcpyAsync(j->dofkeys, init, sizeof init, hipMemcpyHostToDevice, j->ctx->stream));
  HIPCHK(hipMemcpyAsync(j->dofkeys + 4, init, sizeof init, hipMemcpyHostToDevice, j->ctx->stream));
  return ICS_OK;
}

static int check_params(ics_rl* j, const ics_rl_params* p) {
  if (!p) return fail(ICS_EINVAL, "params is NULL");
  if (p->struct_size != sizeof(ics_rl_params))   // first of all: nothing else of *p may be trusted otherwise
    return fail(ICS_EINVAL, "ics_rl_params.struct_size = %u but this library (ABI %d) expects %zu: the caller was built against another include/ics_hip.h",
                p->struct_size, ICS_ABI_VERSION, sizeof(ics_rl_params));
  if (!j) return fail(ICS_EINVAL, "job is NULL");
  if (p->tv_mode < ICS_TV_SHIPPED || p->tv_mode > ICS_TV_PAM_COLLAB)
    return fail(ICS_ENOSUP, "tv_mode %d not implemented (0 shipped, 1 active MM-TV, 2 PAM isotropic, 3 PAM collaborative)", p->tv_mode);
  if (p->tv_mode != ICS_TV_SHIPPED && p->fuse) return fail(ICS_ENOSUP, "fuse = 1 is only available with ICS_TV_SHIPPED");
  if (p->conv < ICS_CONV_AUTO || p->conv > ICS_CONV_FFT) return fail(ICS_EINVAL, "conv = %d is not an ICS_CONV_* value", p->conv);
  if (p->conv == ICS_CONV_FFT && (!(ics_conv_fft_supported(j->g.K) || (ics_conv_fft_blk_supported(j->g.K) && p->tv_mode == ICS_TV_SHIPPED)) || p->tv_mode == ICS_TV_MM_ACTIVE || p->fuse))
    return fail(ICS_ENOSUP, "ICS_CONV_FFT: the transform-tile pipeline is built for PSF sizes 3 ... 85 (the shipped loop and the PAM kinds: tv_mode 0, 2, 3; fuse 0) and, as tap blocks, 87 ... 255 (the shipped loop)");
  if (p->conv == ICS_CONV_FFT && !fft_mirror_fits(j->g))
    return fail(ICS_ENOSUP, "ICS_CONV_FFT: the channel-planar mirror of a %d x %d frame with a %d x %d PSF is %zu bytes, beyond the 2^31 - 1 the tile kernels address "
                "(ICS_CONV_AUTO runs such a frame on the matrix cores)", j->g.M, j->g.N, j->g.K, j->g.K, ics_planar_floats(j->g) * sizeof(float));
  if (p->conv == ICS_CONV_MATRIX && !j->bt_conv && !j->blk_conv) return fail(ICS_ENOSUP, "ICS_CONV_MATRIX: no matrix-core path for this PSF size");
  if (p->conv == ICS_CONV_MATRIX && !j->bt_conv && p->tv_mode != ICS_TV_SHIPPED)   // (the tap-block path has no TV epilogue; never run the fp32 kernels under an explicit MATRIX request)
    return fail(ICS_ENOSUP, "ICS_CONV_MATRIX with tv_mode %d: PSF sizes above 49 run on the matrix cores as tap blocks, which exist for the shipped loop only", p->tv_mode);
  if (p->conv == ICS_CONV_VECTOR && psf_blocks_only(j->g.K))
    return fail(ICS_ENOSUP, "ICS_CONV_VECTOR: PSF sizes above 127 only run as tap blocks on the matrix cores (ICS_CONV_AUTO / ICS_CONV_MATRIX)");
  if (p->fuse && j->g.K > 31) return fail(ICS_ENOSUP, "fuse = 1 is only built for PSF sizes <= 31");
  if (p->tv_mode != ICS_TV_SHIPPED && j->g.K > 63) return fail(ICS_ENOSUP, "tv_mode %d is only built for PSF sizes <= 63 (the shipped loop runs to 255)", p->tv_mode);
  if (p->blind && p->channels != 3) return fail(ICS_ENOSUP, "blind deconvolution requires C == 3 (pyx:557,570 leave gradk undefined otherwise)");
  return ICS_OK;
}

// Small frames (ics_small.hip): every (tile, channel) of the u-frame on a compute unit of its own, the operands of the five inner iterations
// resident in LDS, ONE cooperative launch per outer iteration instead of 25 - 30.  What ICS_CONV_AUTO picks for the shipped loop when the
// frame is small enough (3 x tiles of 32 or 64 <= compute units, operands <= 160 KB of LDS: up to ~290^2 at 31 x 31, ~570^2 at 15 x 15);
// an explicit ICS_CONV_* request, ICS_CONV_PATH or the debug switch `small_iter` = 0 keep the multi-launch families.
static bool use_small_iter(const ics_rl* j, const ics_rl_params* p, IcsSmallPlan* plan = nullptr) {
  if (p->conv != ICS_CONV_AUTO || p->tv_mode != ICS_TV_SHIPPED || p->fuse || j->fft_on || j->small_off) return false;
  if (ics_debug().conv_path.load(std::memory_order_relaxed) != 0 || !ics_debug().small_iter.load(std::memory_order_relaxed)) return false;
  // non-blind (two convolutions and two barriers per inner iteration): measured level with the multi-launch path at 255^2 for PSF sizes up to 15 (0.029 ms
  // either way), ahead of it from 17 x 17 (255^2 / 23: 0.057 -> 0.041) and on frames up to ~160^2 at every size (128^2 / 7: 0.022 -> 0.018)
  if (!p->blind && j->g.K < 17 && (long)j->g.uM * j->g.uN > 26000L) return false;
  IcsSmallPlan pl;
  if (!ics_small_plan(j->g, j->ctx ? j->ctx->cus : 256, &pl, ics_debug().small_iter.load(std::memory_order_relaxed) == 2)) return false;
  if (plan) *plan = pl;
  return true;
}

static int ensure_small(ics_rl* j, const IcsSmallPlan& pl) {
  if (!j->small_part) RC(dalloc(j->ctx, &j->small_part, (size_t)pl.nwg * j->g.K * j->g.K));
  if (!j->small_bar) RC(dalloc(j->ctx, &j->small_bar, (size_t)ICS_SMALL_BAR_WORDS));
  if (!j->small_keys) RC(dalloc(j->ctx, &j->small_keys, (size_t)8 * pl.nwg));
  return ICS_OK;
}

// pyx:462-589 for one outer iteration: ut = u (the untouched input frame is the majoriser), INNER inner iterations, u2 receives u
static int do_small_iter(ics_rl* j, const ics_rl_params* p, const IcsSmallPlan& pl, int inner, Prof& pr) {
  IcsSmallArgs a;
  memset(&a, 0, sizeof a);
  a.u_in = org(j, j->u); a.u_out = org(j, j->u2); a.f = org(j, j->f); a.e = org(j, j->e);
  a.red = red_of(j); a.dofkeys = dof_of(j); a.scal = j->scal;
  a.psf = j->psf; a.psf_caller = j->psf_caller; a.frozen = j->flags;
  a.psf_bak = j->small_bak ? j->psf_bak : nullptr;
  a.part = j->small_part; a.keys = j->small_keys; a.gradk = j->gradk; a.bar = j->small_bar; a.bar_gen = j->small_gen;
  a.step = p->step_factor; a.lambd = p->lambd; a.blind = p->blind; a.correlation = p->correlation; a.inner = inner;
  a.plan = pl; a.g = j->g;
  const bool trace = ics_debug().small_trace.load(std::memory_order_relaxed) != 0;
  unsigned long long* tr = nullptr;
  if (trace && dalloc(j->ctx, &tr, (size_t)pl.nwg * 64) == ICS_OK) a.trace = tr;
  RC(pr.begin(ICS_K_SMALL_ITER));
  const hipError_t he = ics_debug().fail_small_launch.exchange(0) ? hipErrorCooperativeLaunchTooLarge : ics_launch_small_iter(a, j->ctx->stream);
  RC(pr.end());
  if (he != hipSuccess) {   // (nothing was queued: the bracket does not count as a launch)
    if (pr.on && !j->ev_pairs.empty()) j->ev_pairs.pop_back();
    return fail(ICS_EHIP, "cooperative launch of the small-frame iteration: %s", hipGetErrorString(he));
  }
  if (tr) {   // phase timeline: per stamp the first and the last workgroup to reach it, in us from the first stamp of the launch (100 MHz clock)
    std::vector<unsigned long long> h((size_t)pl.nwg * 64);
    HIPCHK(hipStreamSynchronize(j->ctx->stream));
    HIPCHK(hipMemcpy(h.data(), tr, h.size() * 8, hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ull;
    for (int w = 0; w < pl.nwg; ++w) if (h[(size_t)w * 64] && h[(size_t)w * 64] < t0) t0 = h[(size_t)w * 64];
    fprintf(stderr, "small_trace: %d workgroups, stamp: first / last workgroup (us)\n", pl.nwg);
    fprintf(stderr, "  workgroup 0: %llu shader clocks in %.2f us = %.0f MHz\n", h[62], (double)h[63] / 100.0, h[63] ? (double)h[62] / ((double)h[63] / 100.0) : 0.0);
    for (int k = 0; k < 62; ++k) {
      unsigned long long lo = ~0ull, hi = 0;
      for (int w = 0; w < pl.nwg; ++w) { const unsigned long long v = h[(size_t)w * 64 + k]; if (!v) continue; lo = v < lo ? v : lo; hi = v > hi ? v : hi; }
      if (!hi) break;
      fprintf(stderr, "  %2d  %7.2f  %7.2f\n", k, (double)(lo - t0) / 100.0, (double)(hi - t0) / 100.0);
    }
    j->ctx->pool.release(tr);
  }
  j->small_gen += (unsigned long long)ics_small_barriers(p->blind, inner);
  j->small_bak = false;
  // as after the first update of an outer iteration (do_update): the untouched old u is the majoriser, the spare holds u
  float* old_ut = j->ut;
  j->ut = j->u; j->u = j->u2; j->u2 = old_ut;
  j->ut_is_u = false;
  return ICS_OK;
}

// One submission per outer iteration (round 4).  deblur_module runs its blind phase on a 255-px window at every pyramid level
// (deconvolve.py:138-141,277-286): 15 ... 35 launches of 5 ... 15 us each per outer iteration, where the host's launch calls and
// the gaps between dependent dispatches weigh as much as the kernels.  The launches of an outer iteration (pyx:462-638: five inner
// iterations, the statistics, the copy of the scalars to the pinned host mirror) are captured once per frame rotation and replayed
// with hipGraphLaunch.  Not with profiling (events between the kernels), not with the opt-in fused update + convolution (its own
// ping-pong), not with an empty window (host-side NaN upload).  Default: frames up to 1.2 Mpx; debug switch `graph` = 0 / 1 forces.
static bool use_graph(const ics_rl* j, const ics_rl_params* p) {
  if (p->profile || p->fuse || j->win_empty || use_fft_pipeline(j, p, true) || use_small_iter(j, p)) return false;
  const int g = ics_debug().graph.load(std::memory_order_relaxed);
  if (g >= 0) return g != 0;
  return (long)j->g.uM * j->g.uN <= 1200000L;
}

// Statistics of outer iteration i overlapped with iteration i + 1 (ics_rl_run).  Not with the opt-in fused update + convolution (its
// own u ping-pong), not with tv_mode 1 (the image is stepped too: nothing to fall back on), not with an empty window, not for a
// single iteration.  Debug switch `overlap` = 0 restores the drain at every outer boundary.
// Measured (MI355X, ms per inner iteration, drained -> overlapped): non-blind 512^2 / 9x9 0.044 -> 0.040 (0.060 -> 0.040 after a long run of
// small launches, when the device has clocked down), non-blind 2048^2 / 15x15 0.160 -> 0.152; blind 4096^2 / 15x15 0.808 -> 0.804 without
// and 0.810 -> 0.821 with event brackets in the timed region: on frames that fill the device the statistics' small kernels only compete
// with the persistent workgroups of the convolutions.
// Re-measured at the end of round 4 on two boxes, drained -> overlapped: non-blind 512^2 0.045 -> 0.040, 1024^2 0.080 -> 0.073, 1448^2 0.103 -> 0.096,
// 2048^2 level, 2560^2 0.218 -> 0.223, 2900^2 0.270 -> 0.277, 3072^2 0.320 -> 0.333; blind 255^2 0.066 -> 0.066 ... 0.070 (the copies of the PSF
// and the event bubble of an outer boundary weigh as much as the drain they replace), 512^2 0.073 -> 0.077, 1024^2 0.114 -> 0.111, 1536^2
// level, 2048^2 0.227 -> 0.232, 2560^2 0.333 -> 0.341.  Hence by default (switch = 1): non-blind up to 4.5 Mpx, blind between 0.6 and 2.5 Mpx;
// everywhere with switch = 2.
// Switch = 3 (opt-in): the same loop with the statistics queued on the job's OWN stream: no concurrency, but iteration i + 1 is still queued
// before M_r(i) reaches the host, so the device does not idle through the host's round trip at an outer boundary.  Measured: never slower,
// 0 ... 1.5 % faster (blind 4096^2 0.8145 -> 0.8075 ms, blind 512^2 0.075 -> 0.072, 255^2 0.066 -> 0.065, the rest level) -- less than the
// outer iteration a run wastes when its stop test fires (1 / n of a run that stops after n), and a frame more per job: not the default.
// Returns 0: drain at every outer boundary, 1: statistics on the second stream, 2: look-ahead on the job's stream.
#ifndef ICS_LOOKAHEAD_DEFAULT
#define ICS_LOOKAHEAD_DEFAULT 0
#endif
static int use_overlap(const ics_rl* j, const ics_rl_params* p) {
  if (p->fuse || p->tv_mode == ICS_TV_MM_ACTIVE || j->win_empty || p->iterations < 2) return 0;
  const int sw = ics_debug().overlap.load(std::memory_order_relaxed);
  const long px = (long)j->g.uM * j->g.uN;
  if (sw == 2) return 1;
  if (sw == 3) return 2;
  if (sw != 1) return 0;
  // (round 6: on the transform tiles -- one persistent 1024-thread workgroup per CU -- the statistics' small kernels fit beside the iteration's at
  //  every size: 4096^2 / 15, drained -> second stream: blind 0.6713 -> 0.6591 ms, non-blind 0.4734 -> 0.4629; look-ahead on the job's own stream 0.669 / 0.470)
  if (j->fft_on || use_small_iter(j, p)) return 1;   // (the cooperative iteration kernel of small frames leaves room beside it as well: blind 255^2 / 15 0.060 -> 0.052 ms)
  if (p->blind ? (px >= 600000L && px <= 2500000L) : px <= 4500000L) return 1;
  return ICS_LOOKAHEAD_DEFAULT ? 2 : 0;
}

// ics_rl_describe / ics_describe: the routing predicates above, as the launches below evaluate them
static int describe_impl(ics_rl* j, const ics_rl_params* p, ics_rl_route* r) {
  if (!r) return fail(ICS_EINVAL, "route is NULL");
  if (r->struct_size != sizeof(ics_rl_route)) return fail(ICS_EINVAL, "ics_rl_route.struct_size = %u, expected %zu", r->struct_size, sizeof(ics_rl_route));
  RC(check_params(j, p));
  memset(r, 0, sizeof *r);
  r->struct_size = sizeof(ics_rl_route);
  const bool fft = use_fft_pipeline(j, p, true);
  struct Flag { ics_rl* j; bool was; ~Flag() { j->fft_on = was; } } flag{j, j->fft_on};   // (the gradient's predicates read it)
  j->fft_on = fft;
  const bool blocks = !fft && use_block_conv(j, p, 0), matrix = !fft && !blocks && use_matrix_conv(j, p);
  const bool small = !fft && use_small_iter(j, p);
  r->conv_family = small ? 6 : (fft ? 5 : (blocks ? 2 : (matrix ? 1 : (use_big_conv(j, p, 0) ? 4 : 3))));
  r->conv_fp16_split = !small && (blocks || matrix);
  if (p->blind && small) r->gradk_family = 8;
  else if (p->blind) {
    const bool fused = !p->fuse && use_fused_gradk(j, p);
    if (fused) r->gradk_family = 1;
    else if (use_fused_fft(j, p)) r->gradk_family = 7;
    else if (use_fft_gradk(j)) r->gradk_family = 6;
    else if (use_split_gradk(j, p)) r->gradk_family = 3;
    else if (ics_big_supported(j->g.K)) r->gradk_family = 5;
    else r->gradk_family = use_matrix_gradk(j, p) ? 2 : 4;
    r->gradk_fp16_split = r->gradk_family <= 3;
  }
  r->image_in_accumulator_order = small ? 0 : (matrix && use_image_acc(p) && ics_conv_mfma_rs(j->g.K, j->g, j->ctx ? j->ctx->cus : 256) != 0) || (r->gradk_family == 1 && use_image_acc(p));   // (no HIP call on this path: ics_describe has no context and assumes an MI355X's 256 CUs)
  r->graph = use_graph(j, p) ? 1 : 0;
  return ICS_OK;
}
extern "C" int ics_rl_describe(ics_rl* j, const ics_rl_params* p, ics_rl_route* r) { return describe_impl(j, p, r); }
// the same for a shape alone: no device, no job (the predicates read the geometry and which weight tables a job of this PSF size owns)
extern "C" unsigned long long ics_rl_frame_bytes(int M, int N, int MK) {
  if (M < 1 || N < 1 || MK < 3 || !(MK & 1) || !psf_supported(MK)) return 0ull;
  return (unsigned long long)ics_frame_floats(ics_make_geom(M, N, MK)) * 4ull;
}

extern "C" int ics_describe(int M, int N, int MK, const ics_rl_params* p, ics_rl_route* r) {
  if (M < 1 || N < 1 || MK < 3 || !(MK & 1)) return fail(ICS_EINVAL, "bad shape: M=%d N=%d MK=%d (MK odd >= 3)", M, N, MK);
  if (!psf_supported(MK)) return fail(ICS_ENOSUP, "PSF size %d not supported (odd sizes 3..%d)", MK, ICS_PSF_MAX);
  ics_rl shell{};
  shell.g = ics_make_geom(M, N, MK);
  float dummy = 0.f;                                               // non-NULL markers only: nothing is dereferenced
  if (ics_conv_mfma_supported(MK)) shell.bt_conv = shell.bt_corr = &dummy;
  if (MK >= 51) shell.blk_conv = shell.blk_corr = &dummy;
  return describe_impl(&shell, p, r);
}

extern "C" int ics_rl_run(ics_rl* j, const ics_rl_params* p, ics_rl_stats* st) {
  if (!st) return fail(ICS_EINVAL, "stats is NULL");
  if (p && p->struct_size == sizeof(ics_rl_params) && st->struct_size != sizeof(ics_rl_stats))
    return fail(ICS_EINVAL, "ics_rl_stats.struct_size = %u but this library (ABI %d) expects %zu", st->struct_size, ICS_ABI_VERSION, sizeof(ics_rl_stats));
  RC(check_params(j, p));
  if (st->trace_cap < 0) return fail(ICS_EINVAL, "ics_rl_stats.trace_cap = %d", st->trace_cap);
  if (!j->uploaded) return fail(ICS_ESTATE, "ics_rl_run before ics_rl_upload");
  HIPCHK(hipSetDevice(j->ctx->device));
  hipStream_t s = j->ctx->stream;
  RC(ensure_window(j, p));
  const bool tv = p->tv_mode != ICS_TV_SHIPPED;
  if (tv) RC(ensure_tv(j));
  // whatever path leaves this function: the statistics' stream is drained before the caller can release or reuse the job's buffers (an
  // error return inside the overlapped loop used to skip the drain), and the FFT pipeline's flag is cleared
  struct Drain { ics_ctx* c; ~Drain() { if (c->stream2) (void)hipStreamSynchronize(c->stream2); } } drain{j->ctx};
  FftScope fft_scope{j};
  if (use_fft_pipeline(j, p, true)) {   // the frames live as channel-planar mirrors for the duration of the run (ics_planar.hip)
    j->fft_on = true;
    int rc_pl = ensure_planar(j);
    // the mirrors are 7 - 8 more frame-sized buffers: when they do not fit and the tiles were AUTO's choice, the run goes on on the HWC
    // kernels, which need nothing more (an explicit ICS_CONV_FFT, and the PAM kinds' explicit request, report the failure)
    if (rc_pl == ICS_ENOMEM && p->conv == ICS_CONV_AUTO) { j->fft_on = false; (void)hipGetLastError(); }
    else if (rc_pl != ICS_OK) return rc_pl;
  }
  if (j->fft_on) {
    RC(to_planar(j, j->u, s));
    if (!j->plf_valid) { RC(to_planar(j, j->f, s)); j->plf_valid = true; }
    fft_scope.back = true;
    // mode 2's image spectra are 1.6 frames more (128 KB per unit): when they do not fit, the run takes A1 and A3 as two kernels instead
    if (use_conv2(j, p) && !j->fspec && dalloc(j->ctx, &j->fspec, ics_conv2_fft_fspec_floats(j->g), false) != ICS_OK) { j->fspec = nullptr; j->conv2_off = true; (void)hipGetLastError(); }
  }
  IcsSmallPlan small_plan;
  bool small = use_small_iter(j, p, &small_plan);
  if (small && ensure_small(j, small_plan) != ICS_OK) { small = false; (void)hipGetLastError(); }   // (AUTO's choice: the multi-launch path needs nothing more)
  if (small) { HIPCHK(hipMemsetAsync(j->small_bar, 0, (size_t)ICS_SMALL_BAR_WORDS * sizeof(unsigned long long), s)); j->small_gen = 0; j->small_bak = false; }
  {  // everything but the caller's in-fields is overwritten
    ics_rl_stats in = *st;
    memset(st, 0, sizeof *st);
    st->struct_size = in.struct_size; st->trace_cap = in.trace_cap;
    st->trace_M_r = in.trace_M_r; st->trace_Hu = in.trace_Hu; st->trace_varu = in.trace_varu; st->trace_dof_min = in.trace_dof_min; st->trace_dof_max = in.trace_dof_max;
  }
  j->ut_is_u = false;
  Prof pr_on{j, p->profile != 0};       // (per-outer kernels are always bracketed when profiling)
  Prof pr_off{j, false};
  Prof& pr = pr_on;
  j->ev_used = 0; j->ev_pairs.clear(); j->ev_chain = -1;
  double ms[ICS_KERNEL_COUNT] = {0};
  int launches[ICS_KERNEL_COUNT] = {0};
  const int INNER = 5;  // pyx:375
  int it = 0, stop = 0, inner_done = 0;
  float M_r = 0.f, M_r_prev = 0.f, Hu = 0.f, varu = 0.f, dmin = 0.f, dmax = 0.f;
  {   // the job's small state in one launch: flags, the tile counters (the kernels re-arm them; an aborted launch must not leak a count), the
      // accumulators of the window statistics (likewise re-armed by their last kernel), both sets of reduction slots and DoF keys (later
      // outer iterations: re-armed on the device by the kernel that writes the scalars)
    IcsRunResetArgs ra;
    ra.flags = j->flags; ra.sched = j->sched; ra.dacc = j->dacc; ra.ukey = j->ukey; ra.red = j->red; ra.nred = 2 * 8 * ICS_RED_STRIDE; ra.dofkeys = j->dofkeys;
    HIPCHK(ics_launch_run_reset(ra, s));
  }
  // the caller's psf array is the local psf when the call starts (pyx:341)
  HIPCHK(hipMemcpyAsync(j->psf_caller, j->psf, (size_t)3 * j->g.K * j->g.K * 4, hipMemcpyDeviceToDevice, s));
  RC(pack_weights(j, 0, 0.f, 0, s));
  HIPCHK(hipEventRecord(j->ev_begin, s));
  // the launches of one outer iteration (pyx:462-591) ...
  auto enqueue_body = [&]() -> int {
    j->ev_chain = -1;
    if (p->fuse) RC(do_majorize(j, pr));                      // pyx:462 (explicit copy only for the fused path)
    else j->ut_is_u = true;                                   // pyx:462 without a copy (see ut_of)
    if (small) {   // one cooperative launch for the INNER inner iterations (ics_small.hip)
      if (do_small_iter(j, p, small_plan, INNER, p->profile > 0 ? pr_on : pr_off) == ICS_OK) { inner_done += INNER; return ICS_OK; }
      small = false; j->small_off = true; (void)hipGetLastError();   // refused by the runtime: this job runs the multi-launch path from here on
      RC(pack_weights(j, 0, 0.f, 0, s));   // (the weight tables of the multi-launch kernels follow the PSF the cooperative launches have stepped so far)
      if (j->small_bak) {   // (the copy of the PSF the refused launch was to take)
        const size_t npsf = (size_t)3 * j->g.K * j->g.K;
        HIPCHK(hipMemcpyAsync(j->psf_bak, j->psf, npsf * 4, hipMemcpyDeviceToDevice, s));
        HIPCHK(hipMemcpyAsync(j->psf_bak + npsf, j->psf_caller, npsf * 4, hipMemcpyDeviceToDevice, s));
        j->small_bak = false;
      }
    }
    const bool fuse = p->fuse != 0;
    const bool fused_gk = p->blind && !fuse && use_fused_gradk(j, p);
    const bool fused_fft = p->blind && !fuse && use_fused_fft(j, p);
    // (blind without the fused A11 + A13 unit: A11 rewrites the whole residual frame, nothing more to do; either way the window is there)
    const bool conv2 = use_conv2(j, p);
    bool have_e = false;  // error already produced by a fused update+synth kernel
    for (int itt = 0; itt < INNER; ++itt) {                   // pyx:473
      const int last = itt == INNER - 1;
      // profile = k: bracket the launches of every k-th inner iteration only (k = 1: all).  Event records
      // between dependent kernels cost ~4 % of a 4096^2 blind iteration, a sample of them does not.
      Prof& pr = (p->profile > 0 && inner_done % p->profile == 0) ? pr_on : pr_off;
      if (conv2) {
        if (tv) RC(do_tvterm(j, p, itt, pr));                 // (PAM kinds: T of u, read by the back-projection's epilogue)
        RC(do_conv2(j, p, itt, pr));                          // A1 + A2 + A3 (+A7) in one unit per tile pair; the residual frame is not written ...
        if (!p->blind && last) RC(do_conv_fft_window(j, p, pr));   // ... so the statistics' window of it is (blind: A11 rewrites it, do_synth_gradk_fft)
      } else {
      if (!have_e) RC(do_conv(j, 0, p, itt, 0, pr));          // A1+A2
      have_e = false;
      if (tv) RC(do_tvterm(j, p, itt, pr));                   // pyx:495-496 (live only in tv_mode 1)
      RC(do_conv(j, 1, p, itt, 0, pr));                       // A3 (+A7)
      }
      if (p->blind) {                                         // pyx:555
        if (fuse) RC(do_conv(j, 2, p, itt, last, pr));        // A5-A10 fused with A11
        else {
          RC(do_update(j, p, itt, last, pr));
          if (fused_gk) RC(do_synth_gradk(j, p, 0, pr));      // A11 + A12 + A13, e' stays on chip
          else if (fused_fft) RC(do_synth_gradk_fft(j, p, 0, pr));   // the same on the transform tiles
          else RC(do_conv(j, 0, p, itt, 0, pr));
        }
        if (fuse || !(fused_gk || fused_fft)) RC(do_gradk(j, p, pr));   // A12+A13
        RC(do_psf(j, p, pr));                                 // A14-A17
      } else if (fuse && !last) {
        RC(do_conv(j, 2, p, itt, 0, pr));                     // A5-A10 fused with A1+A2 of itt+1
        have_e = true;
      } else {
        RC(do_update(j, p, itt, last, pr));                   // A5,A6,A8,A10 (outer boundary: stats need u and e)
      }
      ++inner_done;
    }
    return ICS_OK;
  };
  // ... and its statistics (A18 + A19) with the copy of the scalars to the pinned host mirror `hs`, on stream `st`
  auto enqueue_stats = [&](hipStream_t st, float* hs, Prof& prs) -> int {
    RC(do_stats(j, p, prs, 1, st));
    HIPCHK(hipMemcpyAsync(hs, j->scal, ICS_SC_COUNT * 4, hipMemcpyDeviceToHost, st));
    return ICS_OK;
  };
  auto enqueue_outer = [&]() -> int {
    RC(enqueue_body());
    return enqueue_stats(s, j->h_scal, pr);
  };
  // One hipGraph launch per outer iteration (use_graph): everything captured carries the parameters in its kernel arguments, so the
  // executables live as long as the parameter set (and the debug switches) stay what they were.
  const bool graphs_on = use_graph(j, p);
  if (graphs_on) {
    ics_rl_params sig;
    memset(&sig, 0, sizeof sig);
    sig.top = p->top; sig.bottom = p->bottom; sig.left = p->left; sig.right = p->right; sig.tau = p->tau; sig.step_factor = p->step_factor;
    sig.lambd = p->lambd; sig.blind = p->blind; sig.correlation = p->correlation; sig.channels = p->channels; sig.tv_mode = p->tv_mode;
    sig.stop_test = p->stop_test; sig.conv = p->conv; sig.flags = p->flags;
    const int epoch = g_debug_epoch.load(std::memory_order_relaxed);
    if (!j->graphs.empty() && (memcmp(&sig, &j->graph_sig, sizeof sig) != 0 || epoch != j->graph_epoch)) {
      for (auto& g : j->graphs) hipGraphExecDestroy(g.exec);
      j->graphs.clear();
    }
    j->graph_sig = sig; j->graph_epoch = epoch;
  }
  // what the host does once the scalars of an outer iteration are in `hs`: traces, stop decision (pyx:643-654), progress callback
  auto consume = [&](const float* hs) {
    if (it > 0) M_r_prev = M_r;                               // pyx:623-624
    M_r = p->stop_test ? hs[ICS_SC_MR] : nanf("");
    Hu = hs[ICS_SC_HU]; varu = hs[ICS_SC_VARU];
    dmin = hs[ICS_SC_DOFMIN]; dmax = hs[ICS_SC_DOFMAX];
    if (it < st->trace_cap) {
      if (st->trace_M_r) st->trace_M_r[it] = M_r;
      if (st->trace_Hu) st->trace_Hu[it] = Hu;
      if (st->trace_varu) st->trace_varu[it] = varu;
      if (st->trace_dof_min) st->trace_dof_min[it] = dmin;
      if (st->trace_dof_max) st->trace_dof_max[it] = dmax;
      st->trace_len = it + 1;
    }
    if (it > 1 && p->stop_test == 1) {                        // pyx:643-654 (stop_test 2: evaluate only)
      if (p->blind) { if (M_r > M_r_prev) stop = 1; }
      else { if ((M_r - M_r_prev) / (M_r + M_r_prev) > p->tau) stop = 1; }
    }
    ++it;
    // pyx:593,648,658-659: where the reference prints.  A non-zero return leaves the loop with the state of this outer iteration
    // (deconvolve.py:338-342 keeps the partial result of an interrupted run)
    if (p->progress && p->progress(p->progress_user, it, stop, dmin, dmax, M_r, Hu, varu) != 0 && !stop) stop = 2;
  };
  j->par = 0;
  const int ovl = graphs_on ? 0 : use_overlap(j, p);
  if (ovl) {
    // ---- statistics of iteration i on a second stream, iteration i + 1 already running on the job's stream (round 4) -------------------
    // The stop decision of iteration i needs M_r(i) on the host, so until round 3 the device drained at every outer boundary: five
    // small dependent kernels (0.06 ms) and a round trip with nothing else in flight -- a quarter of a 512^2 step, 7 % at 2048^2.
    // Now iteration i + 1 is queued before M_r(i) is known.  What that needs: iteration i + 1 may not touch what the statistics of i
    // read or re-arm -- the residual frame ping-pongs (e / e2), the reduction slots and DoF keys come in two sets (i & 1), u(i) is
    // the untouched majoriser of i + 1 anyway (frame rotation) -- and, if the stop test fires at i (or the callback asks to stop),
    // iteration i + 1 is undone: its u is dropped for the majoriser frame (= u(i)), the PSF comes back from the copy taken when i + 1
    // started.  At most one outer iteration is ever ahead, and only the run's last decision costs a wasted one.
    ics_ctx* c = j->ctx;
    if (ovl == 1 && !c->stream2) {   // lowest priority: the statistics take what the iteration's kernels leave free
      int lo = 0, hi = 0;
      HIPCHK(hipDeviceGetStreamPriorityRange(&lo, &hi));
      HIPCHK(hipStreamCreateWithPriority(&c->stream2, hipStreamNonBlocking, lo));
    }
    if (!j->ev_body[0])
      for (int i = 0; i < 2; ++i) { HIPCHK(hipEventCreateWithFlags(&j->ev_body[i], hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&j->ev_stats[i], hipEventDisableTiming)); }
    if (!j->e2) RC(dalloc(c, &j->e2, j->frame_floats));
    if (j->fft_on) RC(ensure_planar(j));                     // (a mirror for e2 as well)
    const size_t npsf = (size_t)3 * j->g.K * j->g.K;
    if (p->blind && !j->psf_bak) RC(dalloc(c, &j->psf_bak, 2 * npsf, false));
    const hipStream_t st2 = ovl == 1 ? c->stream2 : s;      // where the statistics run
    Prof pr_s2{j, p->profile != 0, st2};
    size_t ev_mark[2] = {0, 0}, ev_done = 0;
    int enq = 0;                                              // outer iterations queued; `it` = outer iterations whose scalars were consumed
    auto undo = [&]() -> int {                                // drops iteration enq - 1 (queued, possibly running)
      HIPCHK(hipStreamSynchronize(s));
      if (st2 != s) HIPCHK(hipStreamSynchronize(st2));
      { float* t = j->u; j->u = j->ut; j->ut = t; }           // the majoriser frame of the dropped iteration is u of the one before
      { float* t = j->e; j->e = j->e2; j->e2 = t; }
      if (p->blind) {
        HIPCHK(hipMemcpyAsync(j->psf, j->psf_bak, npsf * 4, hipMemcpyDeviceToDevice, s));
        HIPCHK(hipMemcpyAsync(j->psf_caller, j->psf_bak + npsf, npsf * 4, hipMemcpyDeviceToDevice, s));
        RC(pack_weights(j, 0, 0.f, 0, s));
      }
      inner_done -= INNER;
      // the dropped iteration's event brackets do not enter ms_kernel[] / launches[] (they describe iterations that count); its device time
      // stays inside ms_total, which is the span of the whole call
      if (enq >= 2 && j->ev_pairs.size() > ev_mark[(enq - 2) & 1]) j->ev_pairs.resize(ev_mark[(enq - 2) & 1]);
      return ICS_OK;
    };
    while (enq < p->iterations && !stop) {                    // pyx:460
      j->par = enq & 1;
      if (enq > 0) {
        { float* t = j->e; j->e = j->e2; j->e2 = t; }         // the statistics of the previous iteration read the other frame
        if (p->blind && small) j->small_bak = true;           // (the cooperative kernel takes the copy itself: two launches and a queue switch less per outer iteration)
        else if (p->blind) {
          HIPCHK(hipMemcpyAsync(j->psf_bak, j->psf, npsf * 4, hipMemcpyDeviceToDevice, s));
          HIPCHK(hipMemcpyAsync(j->psf_bak + npsf, j->psf_caller, npsf * 4, hipMemcpyDeviceToDevice, s));
        }
      }
      RC(enqueue_body());
      if (st2 != s) {
        HIPCHK(hipEventRecord(j->ev_body[enq & 1], s));
        HIPCHK(hipStreamWaitEvent(st2, j->ev_body[enq & 1], 0));
      }
      RC(enqueue_stats(st2, j->h_scal + (enq & 1) * (ICS_SC_COUNT + 4), pr_s2));
      HIPCHK(hipEventRecord(j->ev_stats[enq & 1], st2));
      ev_mark[enq & 1] = j->ev_pairs.size();
      j->ev_chain = -1;
      ++enq;
      if (enq >= 2) {                                         // the scalars of the iteration BEFORE the one just queued
        HIPCHK(hipEventSynchronize(j->ev_stats[(enq - 2) & 1]));
        RC(pr.collect_range(ms, launches, ev_done, ev_mark[(enq - 2) & 1]));
        consume(j->h_scal + ((enq - 2) & 1) * (ICS_SC_COUNT + 4));
        if (stop) RC(undo());
      }
    }
    if (!stop && enq > it) {                                  // the last iteration's scalars (nothing is ahead of it)
      HIPCHK(hipEventSynchronize(j->ev_stats[(enq - 1) & 1]));
      consume(j->h_scal + ((enq - 1) & 1) * (ICS_SC_COUNT + 4));
    }
    HIPCHK(hipStreamSynchronize(s));
    if (st2 != s) HIPCHK(hipStreamSynchronize(st2));
    RC(pr.collect_range(ms, launches, ev_done, j->ev_pairs.size()));
    j->ev_used = 0; j->ev_pairs.clear(); j->ev_chain = -1;
    j->par = 0;
  } else
  while (it < p->iterations && !stop) {                       // pyx:460
    if (j->win_empty && it > 0) {   // (no statistics kernel re-arms them for an empty window)
      HIPCHK(hipMemsetAsync(j->red, 0, 8 * ICS_RED_STRIDE * sizeof(uint32_t), s));
      RC(reset_dofkeys(j));
    }
    if (graphs_on && it > 0) {   // (the first outer iteration of a call runs eagerly: first launches configure kernels and build lazily-allocated copies)
      hipGraphExec_t exec = nullptr;
      for (auto& g : j->graphs) if (g.u == j->u && g.ut == j->ut && g.u2 == j->u2) exec = g.exec;
      if (exec) {
        HIPCHK(hipGraphLaunch(exec, s));
        float* old_ut = j->ut;                                // the rotation the captured do_update performed (period 3)
        j->ut = j->u; j->u = j->u2; j->u2 = old_ut; j->ut_is_u = false;
        inner_done += INNER;
      } else {
        ics_rl::Graph g{j->u, j->ut, j->u2, nullptr};
        HIPCHK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        const int rc_body = enqueue_outer();
        hipGraph_t graph = nullptr;
        const hipError_t e_end = hipStreamEndCapture(s, &graph);   // always: the stream must leave capture mode whatever the body returned
        if (rc_body != ICS_OK) { if (graph) hipGraphDestroy(graph); return rc_body; }
        if (e_end != hipSuccess) return fail(ICS_EHIP, "hipStreamEndCapture: %s", hipGetErrorString(e_end));
        const hipError_t e_inst = hipGraphInstantiate(&g.exec, graph, nullptr, nullptr, 0);
        hipGraphDestroy(graph);
        if (e_inst != hipSuccess) return fail(ICS_EHIP, "hipGraphInstantiate: %s", hipGetErrorString(e_inst));
        j->graphs.push_back(g);
        HIPCHK(hipGraphLaunch(g.exec, s));
      }
    } else RC(enqueue_outer());
    // (the wait's wake-up is not what the device idles on here: an event before the statistics + polling through them measured the
    //  same iteration time, 0.8178 vs 0.8185 ms at 4096^2 and 0.1573 vs 0.1562 at 2048^2 non-blind)
    HIPCHK(hipStreamSynchronize(s));
    RC(pr.collect(ms, launches));
    consume(j->h_scal);
  }
  if (j->fft_on) { RC(from_planar(j, j->u, s)); RC(from_planar(j, j->e, s)); fft_scope.back = false; }   // the HWC frames are the job's state between calls
  HIPCHK(ics_launch_hasnan(org(j, j->u), j->g, j->flags + 1, s));
  HIPCHK(hipEventRecord(j->ev_end, s));
  int hflags[4] = {0, 0, 0, 0};
  HIPCHK(hipMemcpyAsync(hflags, j->flags, sizeof hflags, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  float total = 0.f;
  HIPCHK(hipEventElapsedTime(&total, j->ev_begin, j->ev_end));
  st->iterations_done = it; st->stopped = stop; st->has_nan = hflags[1];
  st->M_r = M_r; st->Hu = Hu; st->varu = varu; st->dof_min = dmin; st->dof_max = dmax;
  st->ms_total = total; st->inner_iterations = inner_done;
  for (int k = 0; k < ICS_KERNEL_COUNT; ++k) {
    st->launches[k] = launches[k];
    st->ms_kernel[k] = launches[k] ? (float)(ms[k] / launches[k]) : 0.f;
  }
  return ICS_OK;
}

extern "C" int ics_rl_stage(ics_rl* j, int stage, const ics_rl_params* p) {
  RC(check_params(j, p));
  if (p->conv == ICS_CONV_FFT && p->tv_mode != ICS_TV_SHIPPED)
    return fail(ICS_ENOSUP, "ICS_CONV_FFT with tv_mode %d: single stages of the PAM kinds run on the HWC kernels (the tiles serve ics_rl_run)", p->tv_mode);
  j->par = 0;
  HIPCHK(hipSetDevice(j->ctx->device));
  hipStream_t s = j->ctx->stream;
  Prof pr{j, false};
  HIPCHK(hipMemsetAsync(j->sched, 0, 16 * sizeof(uint32_t), s));
  // conv = ICS_CONV_FFT through the stage API (tests): the stage runs on the mirrors, every frame is copied over before and the frames a
  // stage writes are copied back after -- slow and simple; ics_rl_run converts at its boundaries only
  FftScope fft_scope{j};
  const bool fft_stage = use_fft_pipeline(j, p, false) && (stage == ICS_STAGE_SYNTH_RESIDUAL || stage == ICS_STAGE_BACKPROJECT || stage == ICS_STAGE_UPDATE ||
                                                            stage == ICS_STAGE_PSF_GRADIENT || stage == ICS_STAGE_PSF_UPDATE || stage == ICS_STAGE_SYNTH_GRADK || stage == ICS_STAGE_SYNTH_BACKPROJECT);
  if (fft_stage) {
    j->fft_on = true;
    RC(ensure_planar(j));
    for (float* h : {j->u, j->ut, j->gr, j->f, j->e}) RC(to_planar(j, h, s));
    j->plf_valid = true;
  }
  struct StageBack {   // (runs before fft_scope clears the flag: declared after it)
    ics_rl* j; bool on; hipStream_t s;
    ~StageBack() { if (on) { for (float* h : {j->u, j->gr, j->e}) (void)from_planar(j, h, s); (void)hipStreamSynchronize(s); } }
  } stage_back{j, fft_stage, s};
  switch (stage) {
    case ICS_STAGE_SYNTH_RESIDUAL:
      RC(pack_weights(j, 0, 0.f, 0, s));
      RC(do_conv(j, 0, p, 0, 0, pr));
      break;
    case ICS_STAGE_BACKPROJECT:
      RC(pack_weights(j, 0, 0.f, 0, s));
      // (in tv_mode 1 ICS_STAGE_TVTERM runs first and owns the reset: its keys live in the same slot)
      if (p->tv_mode == ICS_TV_SHIPPED) HIPCHK(hipMemsetAsync(j->red, 0, 8 * ICS_RED_STRIDE * sizeof(uint32_t), s));
      else RC(ensure_tv(j));
      RC(do_conv(j, 1, p, 0, 0, pr));
      break;
    case ICS_STAGE_UPDATE:
      RC(reset_dofkeys(j));
      RC(do_update(j, p, 0, 1, pr));
      break;
    case ICS_STAGE_TVTERM:
      if (p->tv_mode == ICS_TV_SHIPPED) return fail(ICS_EINVAL, "ICS_STAGE_TVTERM needs tv_mode != ICS_TV_SHIPPED");
      RC(ensure_tv(j));
      HIPCHK(hipMemsetAsync(j->red, 0, 8 * ICS_RED_STRIDE * sizeof(uint32_t), s));
      RC(do_tvterm(j, p, 0, pr));
      break;
    case ICS_STAGE_UPDATE_SYNTH:
      RC(pack_weights(j, 0, 0.f, 0, s));
      RC(reset_dofkeys(j));
      RC(do_conv(j, 2, p, 0, 1, pr));
      break;
    case ICS_STAGE_PSF_GRADIENT: RC(do_gradk(j, p, pr)); break;
    case ICS_STAGE_BAND_REDUCE:
      if (p->tv_mode != ICS_TV_SHIPPED) return fail(ICS_ENOSUP, "row bands are built for the shipped loop (tv_mode 0)");
      if (p->band_row0 < 0 || p->band_row1 > j->g.uM || p->band_row0 >= p->band_row1) return fail(ICS_EINVAL, "band rows [%d, %d) outside the %d u rows", p->band_row0, p->band_row1, j->g.uM);
      HIPCHK(hipMemsetAsync(j->red, 0, ICS_RED_STRIDE * sizeof(uint32_t), s));
      HIPCHK(ics_launch_band_reduce(org(j, j->gr), org(j, j->u), org(j, ut_of(j)), j->g, p->lambd, p->band_row0, p->band_row1, j->red, s));
      break;
    case ICS_STAGE_BAND_MASK_E:
      if (p->band_row0 < 0 || p->band_row1 > j->g.M || p->band_row0 > p->band_row1) return fail(ICS_EINVAL, "band rows [%d, %d) outside the %d image rows", p->band_row0, p->band_row1, j->g.M);
      HIPCHK(ics_launch_band_mask_e(org(j, j->e), j->g, p->band_row0, p->band_row1, s));
      break;
    case ICS_STAGE_SYNTH_GRADK:
      if (j->fft_on && !ics_conv_fft_supported(j->g.K)) return fail(ICS_ENOSUP, "ICS_STAGE_SYNTH_GRADK on the tiles: the fused unit is built for PSF sizes that fit one tile (<= 85); above, A11 and A13 run as tap blocks");
      if (j->fft_on) {   // conv = ICS_CONV_FFT: the fused unit of the transform tiles, every tile stores its residual
        RC(pack_weights(j, 0, 0.f, 0, s));
        RC(do_synth_gradk_fft(j, p, 1, pr));
        break;
      }
      if (!ics_synth_gradk_supported(j->g.K) || !j->bt_conv) return fail(ICS_ENOSUP, "ICS_STAGE_SYNTH_GRADK is built for PSF sizes <= 15");
      RC(pack_weights(j, 0, 0.f, 0, s));
      RC(do_synth_gradk(j, p, 1, pr));
      break;
    case ICS_STAGE_SYNTH_BACKPROJECT:
      if (!j->fft_on || p->tv_mode != ICS_TV_SHIPPED || !ics_conv2_fft_supported(j->g))
        return fail(ICS_ENOSUP, "ICS_STAGE_SYNTH_BACKPROJECT needs params.conv = ICS_CONV_FFT, tv_mode 0 and a PSF of at most 57 x 57");
      RC(pack_weights(j, 0, 0.f, 0, s));
      HIPCHK(hipMemsetAsync(j->red, 0, 8 * ICS_RED_STRIDE * sizeof(uint32_t), s));
      RC(do_conv2(j, p, 0, pr));
      break;
    case ICS_STAGE_PSF_UPDATE: RC(do_psf(j, p, pr)); break;
    case ICS_STAGE_MAJORIZE: RC(do_majorize(j, pr)); break;
    case ICS_STAGE_STATS:
      HIPCHK(hipMemsetAsync(j->dacc, 0, 8 * sizeof(double), s));
      HIPCHK(hipMemsetAsync(j->ukey, 0, 2 * sizeof(uint32_t), s));
      RC(ensure_window(j, p));
      RC(do_stats(j, p, pr));
      break;
    default: return fail(ICS_EINVAL, "unknown stage %d", stage);
  }
  if (!(p->flags & ICS_FLAG_STAGE_ASYNC)) HIPCHK(hipStreamSynchronize(s));
  return ICS_OK;
}

// -------------------------------------------------------------------------------------------------
// standalone operators
namespace {
// lib/deconvolution.pyx:47-70: clamp negatives, divide each channel by its sequential float32 sum
__global__ __launch_bounds__(256) void k_normalize(float* kern, int K) {
  __shared__ float ssum[4];
  const int n = 3 * K * K, tid = threadIdx.x;
  for (int i = tid; i < n; i += 256) if (kern[i] < 0.f) kern[i] = 0.f;
  __syncthreads();
  if (tid < 3) {
    float s = 0.f;
    for (int i = 0; i < K * K; ++i) s = __fadd_rn(s, kern[3 * i + tid]);
    ssum[tid] = s;
  }
  __syncthreads();
  for (int i = tid; i < n; i += 256) kern[i] = __fdiv_rn(kern[i], ssum[i % 3]);
}
}  // namespace

extern "C" int ics_normalize_kernel(ics_ctx* c, float* kern, int MK) {
  if (!c || !kern) return fail(ICS_EINVAL, "NULL argument");
  if (MK < 1) return fail(ICS_EINVAL, "MK = %d", MK);
  HIPCHK(hipSetDevice(c->device));
  const size_t n = (size_t)3 * MK * MK;
  float* d = nullptr;
  HIPCHK(c->pool.alloc((void**)&d, n * 4));
  hipError_t e = hipMemcpyAsync(d, kern, n * 4, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) { hipLaunchKernelGGL(k_normalize, dim3(1), dim3(256), 0, c->stream, d, MK); e = hipGetLastError(); }
  if (e == hipSuccess) e = hipMemcpyAsync(kern, d, n * 4, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  c->pool.release(d);
  if (e != hipSuccess) return fail(ICS_EHIP, "normalize_kernel: %s", hipGetErrorString(e));
  return ICS_OK;
}

extern "C" int ics_tv(ics_ctx* c, const float* u, int M, int N, float eps, int order, int norm, float* out, float* div) {
  if (!c || !u || !out || !div) return fail(ICS_EINVAL, "NULL argument");
  if ((order != 1 && order != 2) || (norm != 1 && norm != 2)) return fail(ICS_EINVAL, "order/norm must be 1 or 2");
  if (M < 1 || N < 1) return fail(ICS_EINVAL, "size %dx%d", M, N);
  HIPCHK(hipSetDevice(c->device));
  const size_t n = (size_t)M * N * 3;
  float *du = nullptr, *dout = nullptr, *ddiv = nullptr;
  hipError_t e = c->pool.alloc((void**)&du, n * 4);
  if (e == hipSuccess) e = c->pool.alloc((void**)&dout, n * 4);
  if (e == hipSuccess) e = c->pool.alloc((void**)&ddiv, n * 4);
  if (e == hipSuccess) e = hipMemcpyAsync(du, u, n * 4, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) e = hipMemsetAsync(dout, 0, n * 4, c->stream);
  if (e == hipSuccess) e = hipMemsetAsync(ddiv, 0, n * 4, c->stream);
  if (e == hipSuccess) e = ics_launch_tv(du, M, N, eps, order, norm, dout, ddiv, c->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(out, dout, n * 4, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(div, ddiv, n * 4, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  c->pool.release(du); c->pool.release(dout); c->pool.release(ddiv);
  if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? ICS_ENOMEM : ICS_EHIP, "tv: %s", hipGetErrorString(e));
  return ICS_OK;
}

// Rank-1 test: kern == outer(col, row)?  Every window of lib/utils.py (uniform, gaussian, kaiser, poisson) is an outer product
// normalised by its sum; the two 1-D factors are taken through the largest element.
static bool rank1_factors(const double* k, int KH, int KW, std::vector<double>& col, std::vector<double>& row) {
  int r = 0, c = 0; double m = 0.0;
  for (int i = 0; i < KH; ++i) for (int j = 0; j < KW; ++j) if (fabs(k[i * KW + j]) > m) { m = fabs(k[i * KW + j]); r = i; c = j; }
  if (m == 0.0 || KH == 1 || KW == 1) return false;
  const double piv = k[r * KW + c];
  for (int i = 0; i < KH; ++i)
    for (int j = 0; j < KW; ++j)
      if (fabs(k[i * KW + j] * piv - k[i * KW + c] * k[r * KW + j]) > 4e-16 * m * m) return false;
  col.resize(KH); row.resize(KW);
  for (int i = 0; i < KH; ++i) col[i] = k[i * KW + c] / piv;
  for (int j = 0; j < KW; ++j) row[j] = k[r * KW + j];
  return true;
}

static int conv2d_common(ics_ctx* c, const double* src, int H, int W, const double* kern, int KH, int KW, int usm, double amount, double* out) {
  if (!c || !src || !kern || !out) return fail(ICS_EINVAL, "NULL argument");
  if (H < 1 || W < 1 || KH < 1 || KW < 1) return fail(ICS_EINVAL, "bad sizes");
  if ((size_t)(32 + KH - 1) * (32 + KW - 1) * 8 > 160 * 1024) return fail(ICS_ENOSUP, "kernel %d x %d too large for the LDS tile (up to 111 x 111)", KH, KW);
  HIPCHK(hipSetDevice(c->device));
  const size_t n = (size_t)H * W, nk = (size_t)KH * KW;
  std::vector<double> col, row;
  const bool sep = rank1_factors(kern, KH, KW, col, row);
  void* base = nullptr;
  int rc = ctx_scratch(c, (3 * n + nk + KH + KW + 16) * 8, &base);
  if (rc != ICS_OK) return fail(rc, "device scratch of %zu bytes", (3 * n + nk) * 8);
  double *ds = (double*)base, *dout = ds + n, *dtmp = dout + n, *dk = dtmp + n;
  hipStream_t s = c->stream;
  HIPCHK(hipMemcpyAsync(ds, src, n * 8, hipMemcpyHostToDevice, s));
  if (sep) {
    HIPCHK(hipMemcpyAsync(dk, row.data(), KW * 8, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(dk + KW, col.data(), KH * 8, hipMemcpyHostToDevice, s));
  } else {
    HIPCHK(hipMemcpyAsync(dk, kern, nk * 8, hipMemcpyHostToDevice, s));
  }
  HIPCHK(hipEventRecord(c->ev0, s));
  if (sep) {   // rows (1 x KW), then columns (KH x 1) with the USM epilogue against the original channel
    HIPCHK(ics_launch_conv2d_symm(ds, H, W, dk, 1, KW, dtmp, ds, 0, 0.0, s));
    HIPCHK(ics_launch_conv2d_symm(dtmp, H, W, dk + KW, KH, 1, dout, ds, usm, amount, s));
  } else {
    HIPCHK(ics_launch_conv2d_symm(ds, H, W, dk, KH, KW, dout, ds, usm, amount, s));
  }
  HIPCHK(hipEventRecord(c->ev1, s));
  HIPCHK(hipMemcpyAsync(out, dout, n * 8, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));   // (col / row are host vectors read by the async copies above)
  HIPCHK(hipEventElapsedTime(&c->last_ms, c->ev0, c->ev1));
  return ICS_OK;
}

extern "C" int ics_conv2d_symm(ics_ctx* c, const double* src, int H, int W, const double* kern, int KH, int KW, double* out) {
  return conv2d_common(c, src, H, W, kern, KH, KW, 0, 0.0, out);
}
extern "C" int ics_usm(ics_ctx* c, const double* src, int H, int W, const double* kern, int KH, int KW, double amount, double* out) {
  return conv2d_common(c, src, H, W, kern, KH, KW, 1, amount, out);
}

extern "C" int ics_bilateral(ics_ctx* c, const double* src, int H, int W, int radius, double std_i, double std_s, double* out) {
  if (!c || !src || !out) return fail(ICS_EINVAL, "NULL argument");
  if (H < 1 || W < 1 || radius < 0) return fail(ICS_EINVAL, "bad sizes");
  if ((size_t)(32 + 2 * radius) * (32 + 2 * radius) * 8 > 160 * 1024) return fail(ICS_ENOSUP, "radius %d too large for the LDS tile (up to 55)", radius);
  HIPCHK(hipSetDevice(c->device));
  const size_t n = (size_t)H * W;
  const int D = 2 * radius + 1;
  std::vector<double> ws((size_t)D * D);
  for (int j = -radius; j <= radius; ++j)
    for (int i = -radius; i <= radius; ++i) ws[(size_t)(j + radius) * D + (i + radius)] = exp((double)(i * i + j * j) * (-1.0 / (2.0 * std_s * std_s)));
  void* base = nullptr;
  int rc = ctx_scratch(c, (2 * n + ws.size() + 16) * 8, &base);
  if (rc != ICS_OK) return fail(rc, "device scratch of %zu bytes", 2 * n * 8);
  double *ds = (double*)base, *dout = ds + n, *dws = dout + n;
  hipStream_t s = c->stream;
  HIPCHK(hipMemcpyAsync(ds, src, n * 8, hipMemcpyHostToDevice, s));
  HIPCHK(hipMemcpyAsync(dws, ws.data(), ws.size() * 8, hipMemcpyHostToDevice, s));
  HIPCHK(hipEventRecord(c->ev0, s));
  HIPCHK(ics_launch_bilateral(ds, H, W, radius, std_i, dws, dout, s));
  HIPCHK(hipEventRecord(c->ev1, s));
  HIPCHK(hipMemcpyAsync(out, dout, n * 8, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  HIPCHK(hipEventElapsedTime(&c->last_ms, c->ev0, c->ev1));
  return ICS_OK;
}

// deconvolve.py:245-249 -- skimage.transform.resize(order=3, mode="edge") restated on scipy.ndimage semantics (oracle/resize_oracle.py)
extern "C" int ics_resize_bicubic(ics_ctx* c, const double* src, int H, int W, int C, double* out, int OH, int OW) {
  if (!c || !src || !out) return fail(ICS_EINVAL, "NULL argument");
  if (H < 2 || W < 2 || C < 1 || OH < 1 || OW < 1) return fail(ICS_EINVAL, "bad sizes");
  HIPCHK(hipSetDevice(c->device));
  const size_t n = (size_t)H * W * C, no = (size_t)OH * OW * C;
  // Gaussian anti-aliasing weights (host, float64 like scipy.ndimage.gaussian_filter1d)
  auto weights = [](double sigma, std::vector<double>& w) {
    const int r = (int)(4.0 * sigma + 0.5);
    w.resize(2 * r + 1);
    double sum = 0.0;
    for (int k = -r; k <= r; ++k) { w[k + r] = exp(-0.5 / (sigma * sigma) * (double)k * (double)k); sum += w[k + r]; }
    for (double& v : w) v /= sum;
    return r;
  };
  const double sy = fmax(0.0, ((double)H / OH - 1.0) / 2.0), sx = fmax(0.0, ((double)W / OW - 1.0) / 2.0);
  const bool smooth = (sy > 0.0 || sx > 0.0) && !(H == OH && W == OW);
  std::vector<double> hwy, hwx;
  int ry = 0, rx = 0;
  if (smooth && sy > 1e-15) ry = weights(sy, hwy);
  if (smooth && sx > 1e-15) rx = weights(sx, hwx);
  double *ds = nullptr, *scr = nullptr, *dout = nullptr, *dw = nullptr;
  hipError_t e = c->pool.alloc((void**)&ds, n * 8);
  if (e == hipSuccess) e = c->pool.alloc((void**)&scr, ics_resize_scratch_doubles(H, W, C) * 8);
  if (e == hipSuccess) e = c->pool.alloc((void**)&dout, no * 8);
  if (e == hipSuccess) e = c->pool.alloc((void**)&dw, (hwy.size() + hwx.size() + 1) * 8);
  if (e == hipSuccess) e = hipMemcpyAsync(ds, src, n * 8, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess && !hwy.empty()) e = hipMemcpyAsync(dw, hwy.data(), hwy.size() * 8, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess && !hwx.empty()) e = hipMemcpyAsync(dw + hwy.size(), hwx.data(), hwx.size() * 8, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) {
    if (H == OH && W == OW) e = hipMemcpyAsync(dout, ds, n * 8, hipMemcpyDeviceToDevice, c->stream);
    else e = ics_launch_resize(ds, H, W, C, hwy.empty() ? nullptr : dw, ry, hwx.empty() ? nullptr : dw + hwy.size(), rx, scr, dout, OH, OW, c->stream);
  }
  if (e == hipSuccess) e = hipMemcpyAsync(out, dout, no * 8, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);   // (also keeps hwy / hwx alive until the copies are done)
  c->pool.release(ds); c->pool.release(scr); c->pool.release(dout); c->pool.release(dw);
  if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? ICS_ENOMEM : ICS_EHIP, "resize: %s", hipGetErrorString(e));
  return ICS_OK;
}

// ================================================================================================
// Device-resident images: the frames deconvolve.py keeps between two richardson_lucy_MM calls (pyramid levels, blind ->
// non-blind phase) stay in HBM; every operation is queued on the context's stream, only ics_img_download synchronises.
// ================================================================================================
struct ics_img {
  ics_ctx* ctx;
  int H, W;
  float* d;
};

static int img_new(ics_ctx* c, int H, int W, ics_img** out) {
  if (!c || !out) return fail(ICS_EINVAL, "NULL argument");
  if (H < 1 || W < 1) return fail(ICS_EINVAL, "bad image size %d x %d", H, W);
  HIPCHK(hipSetDevice(c->device));
  ics_img* m = new (std::nothrow) ics_img{c, H, W, nullptr};
  if (!m) return fail(ICS_ENOMEM, "host allocation failed");
  hipError_t e = c->pool.alloc((void**)&m->d, (size_t)H * W * 3 * 4);
  if (e != hipSuccess) { delete m; return fail(ICS_ENOMEM, "device allocation of a %d x %d image: %s", H, W, hipGetErrorString(e)); }
  *out = m;
  return ICS_OK;
}

extern "C" int ics_img_create(ics_ctx* c, int H, int W, ics_img** out) { return img_new(c, H, W, out); }
extern "C" void ics_img_destroy(ics_img* m) {
  if (!m) return;
  m->ctx->pool.release(m->d);   // (operations on the image are queued on the context's stream; so is whatever reuses the block)
  delete m;
}
extern "C" int ics_img_shape(const ics_img* m, int* H, int* W) {
  if (!m) return fail(ICS_EINVAL, "image is NULL");
  if (H) *H = m->H;
  if (W) *W = m->W;
  return ICS_OK;
}
extern "C" int ics_img_upload(ics_img* m, const float* host) {
  if (!m || !host) return fail(ICS_EINVAL, "NULL argument");
  HIPCHK(hipSetDevice(m->ctx->device));
  HIPCHK(hipMemcpyAsync(m->d, host, (size_t)m->H * m->W * 12, hipMemcpyHostToDevice, m->ctx->stream));
  HIPCHK(hipStreamSynchronize(m->ctx->stream));   // the host buffer may be released by the caller
  return ICS_OK;
}
extern "C" int ics_img_upload_int(ics_img* m, const void* host, int bytes_per_value) {
  if (!m || !host) return fail(ICS_EINVAL, "NULL argument");
  if (bytes_per_value != 1 && bytes_per_value != 2) return fail(ICS_EINVAL, "bytes_per_value = %d (1: uint8, 2: uint16)", bytes_per_value);
  ics_ctx* c = m->ctx;
  HIPCHK(hipSetDevice(c->device));
  const size_t n = (size_t)m->H * m->W * 3;
  void* raw = nullptr;
  if (hipError_t e = c->pool.alloc(&raw, n * bytes_per_value); e != hipSuccess) return fail(ICS_ENOMEM, "img_upload_int: %s", hipGetErrorString(e));
  hipError_t e = hipMemcpyAsync(raw, host, n * bytes_per_value, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) e = ics_launch_int_to_f32(raw, bytes_per_value, m->d, (long)n, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);   // the host buffer may be released by the caller
  c->pool.release(raw);                                       // (everything of a context runs on its one stream: a recycled block needs no more)
  if (e != hipSuccess) return fail(ICS_EHIP, "img_upload_int: %s", hipGetErrorString(e));
  return ICS_OK;
}
extern "C" int ics_img_download(const ics_img* m, float* host) {
  if (!m || !host) return fail(ICS_EINVAL, "NULL argument");
  HIPCHK(hipSetDevice(m->ctx->device));
  HIPCHK(hipMemcpyAsync(host, m->d, (size_t)m->H * m->W * 12, hipMemcpyDeviceToHost, m->ctx->stream));
  HIPCHK(hipStreamSynchronize(m->ctx->stream));
  return ICS_OK;
}
extern "C" int ics_img_pad_edge(const ics_img* src, int top, int bottom, int left, int right, ics_img** out) {
  if (!src || !out) return fail(ICS_EINVAL, "NULL argument");
  if (top < 0 || bottom < 0 || left < 0 || right < 0) return fail(ICS_EINVAL, "negative padding");
  RC(img_new(src->ctx, src->H + top + bottom, src->W + left + right, out));
  hipError_t e = ics_launch_img_pad_edge(src->d, src->H, src->W, (*out)->d, top, bottom, left, right, src->ctx->stream);
  if (e != hipSuccess) { ics_img_destroy(*out); *out = nullptr; return fail(ICS_EHIP, "img_pad_edge: %s", hipGetErrorString(e)); }
  return ICS_OK;
}
static int rect_ok(const ics_img* m, int y0, int x0, int H, int W) { return y0 >= 0 && x0 >= 0 && H >= 1 && W >= 1 && y0 + H <= m->H && x0 + W <= m->W; }
extern "C" int ics_img_crop(const ics_img* src, int y0, int x0, int H, int W, ics_img** out) {
  if (!src || !out) return fail(ICS_EINVAL, "NULL argument");
  if (!rect_ok(src, y0, x0, H, W)) return fail(ICS_EINVAL, "crop [%d:%d, %d:%d] outside a %d x %d image", y0, y0 + H, x0, x0 + W, src->H, src->W);
  RC(img_new(src->ctx, H, W, out));
  hipError_t e = hipMemcpy2DAsync((*out)->d, (size_t)W * 12, src->d + ((size_t)y0 * src->W + x0) * 3, (size_t)src->W * 12, (size_t)W * 12, H,
                                  hipMemcpyDeviceToDevice, src->ctx->stream);
  if (e != hipSuccess) { ics_img_destroy(*out); *out = nullptr; return fail(ICS_EHIP, "img_crop: %s", hipGetErrorString(e)); }
  return ICS_OK;
}
extern "C" int ics_img_paste(ics_img* dst, int y0, int x0, const ics_img* src) {
  if (!src || !dst) return fail(ICS_EINVAL, "NULL argument");
  if (src->ctx != dst->ctx) return fail(ICS_EINVAL, "images of different contexts");
  if (!rect_ok(dst, y0, x0, src->H, src->W)) return fail(ICS_EINVAL, "paste of %d x %d at (%d, %d) outside a %d x %d image", src->H, src->W, y0, x0, dst->H, dst->W);
  HIPCHK(hipSetDevice(dst->ctx->device));
  HIPCHK(hipMemcpy2DAsync(dst->d + ((size_t)y0 * dst->W + x0) * 3, (size_t)dst->W * 12, src->d, (size_t)src->W * 12, (size_t)src->W * 12, src->H,
                          hipMemcpyDeviceToDevice, dst->ctx->stream));
  return ICS_OK;
}
extern "C" int ics_img_gamma(ics_img* m, float div, float exponent, float mul, int clip01) {
  if (!m) return fail(ICS_EINVAL, "image is NULL");
  HIPCHK(hipSetDevice(m->ctx->device));
  HIPCHK(ics_launch_img_gamma(m->d, (long)m->H * m->W * 3, div, exponent, mul, clip01, m->ctx->stream));
  return ICS_OK;
}
// deconvolve.py:245-249 on a device image: float64 inside (as skimage / scipy compute), rounded to float32 like the
// reference's `.astype(np.float32)`
extern "C" int ics_img_resize(const ics_img* src, int OH, int OW, ics_img** out) {
  if (!src || !out) return fail(ICS_EINVAL, "NULL argument");
  if (OH < 1 || OW < 1 || src->H < 2 || src->W < 2) return fail(ICS_EINVAL, "bad sizes");
  ics_ctx* c = src->ctx;
  HIPCHK(hipSetDevice(c->device));
  hipStream_t s = c->stream;
  const int H = src->H, W = src->W;
  if (H == OH && W == OW) return ics_img_crop(src, 0, 0, H, W, out);
  RC(img_new(c, OH, OW, out));
  auto weights = [](double sigma, std::vector<double>& w) {
    const int r = (int)(4.0 * sigma + 0.5);
    w.resize(2 * r + 1);
    double sum = 0.0;
    for (int k = -r; k <= r; ++k) { w[k + r] = exp(-0.5 / (sigma * sigma) * (double)k * (double)k); sum += w[k + r]; }
    for (double& v : w) v /= sum;
    return r;
  };
  const double sy = fmax(0.0, ((double)H / OH - 1.0) / 2.0), sx = fmax(0.0, ((double)W / OW - 1.0) / 2.0);
  std::vector<double> hwy, hwx;
  int ry = 0, rx = 0;
  if (sy > 1e-15) ry = weights(sy, hwy);
  if (sx > 1e-15) rx = weights(sx, hwx);
  double *scr = nullptr, *dw = nullptr;          // (the float32 frames are read and written by the float64 pipeline's first and last pass)
  hipError_t e = c->pool.alloc((void**)&scr, ics_resize_scratch_doubles(H, W, 3) * 8);
  if (e == hipSuccess) e = c->pool.alloc((void**)&dw, (hwy.size() + hwx.size() + 1) * 8);
  const size_t nw = hwy.size() + hwx.size();
  bool staged = false;
  if (e == hipSuccess && nw) {
    if (nw <= ics_ctx::PIN_DOUBLES) {          // through the context's pinned staging area: nothing to wait for afterwards
      if (!c->pin) { e = hipHostMalloc((void**)&c->pin, ics_ctx::PIN_DOUBLES * 8, hipHostMallocDefault); if (e == hipSuccess) e = hipEventCreateWithFlags(&c->pin_ev, hipEventDisableTiming); }
      if (e == hipSuccess && c->pin_used) e = hipEventSynchronize(c->pin_ev);
      if (e == hipSuccess) {
        memcpy(c->pin, hwy.data(), hwy.size() * 8);
        memcpy(c->pin + hwy.size(), hwx.data(), hwx.size() * 8);
        e = hipMemcpyAsync(dw, c->pin, nw * 8, hipMemcpyHostToDevice, s);
        if (e == hipSuccess) e = hipEventRecord(c->pin_ev, s);
        c->pin_used = true; staged = true;
      }
    } else {
      if (!hwy.empty()) e = hipMemcpyAsync(dw, hwy.data(), hwy.size() * 8, hipMemcpyHostToDevice, s);
      if (e == hipSuccess && !hwx.empty()) e = hipMemcpyAsync(dw + hwy.size(), hwx.data(), hwx.size() * 8, hipMemcpyHostToDevice, s);
    }
  }
  if (e == hipSuccess) e = ics_launch_resize_f32(src->d, H, W, 3, hwy.empty() ? nullptr : dw, ry, hwx.empty() ? nullptr : dw + hwy.size(), rx, scr, (*out)->d, OH, OW, s);
  if (e == hipSuccess && nw && !staged) e = hipStreamSynchronize(s);   // pageable host vectors are released below
  c->pool.release(scr); c->pool.release(dw);
  if (e != hipSuccess) { ics_img_destroy(*out); *out = nullptr; return fail(e == hipErrorOutOfMemory ? ICS_ENOMEM : ICS_EHIP, "img_resize: %s", hipGetErrorString(e)); }
  return ICS_OK;
}

// richardson_lucy_MM(image[iy:iy+M, ix:ix+N], u[uy:uy+uM, ux:ux+uN], psf, ...) with both arrays on the device
// (deconvolve.py:277-313 passes such window views)
extern "C" int ics_rl_upload_img(ics_rl* j, const ics_img* image, int iy, int ix, const ics_img* u, int uy, int ux, const float* psf) {
  if (!j || !image || !u || !psf) return fail(ICS_EINVAL, "NULL argument");
  const IcsGeom& g = j->g;
  if (image->ctx != j->ctx || u->ctx != j->ctx) return fail(ICS_EINVAL, "images of another context");
  if (!rect_ok(image, iy, ix, g.M, g.N)) return fail(ICS_EINVAL, "image window [%d:%d, %d:%d] outside a %d x %d image", iy, iy + g.M, ix, ix + g.N, image->H, image->W);
  if (!rect_ok(u, uy, ux, g.uM, g.uN)) return fail(ICS_EINVAL, "u window [%d:%d, %d:%d] outside a %d x %d image", uy, uy + g.uM, ux, ux + g.uN, u->H, u->W);
  HIPCHK(hipSetDevice(j->ctx->device));
  hipStream_t s = j->ctx->stream;
  image_changed(j);
  float* df = org(j, j->f) + (ptrdiff_t)g.pad * g.pitch + 3 * g.pad;
  HIPCHK(hipMemcpy2DAsync(df, (size_t)g.pitch * 4, image->d + ((size_t)iy * image->W + ix) * 3, (size_t)image->W * 12, (size_t)g.N * 12, g.M, hipMemcpyDeviceToDevice, s));
  HIPCHK(hipMemcpy2DAsync(org(j, j->u), (size_t)g.pitch * 4, u->d + ((size_t)uy * u->W + ux) * 3, (size_t)u->W * 12, (size_t)g.uN * 12, g.uM, hipMemcpyDeviceToDevice, s));
  const size_t n = (size_t)3 * g.K * g.K * 4;
  HIPCHK(hipMemcpyAsync(j->psf, psf, n, hipMemcpyHostToDevice, s));
  HIPCHK(hipMemcpyAsync(j->psf_caller, psf, n, hipMemcpyHostToDevice, s));
  HIPCHK(hipMemsetAsync(j->flags, 0, 4 * sizeof(int), s));
  RC(pack_weights(j, 0, 0.f, 0, s));
  HIPCHK(hipStreamSynchronize(s));   // psf is a host buffer
  j->uploaded = true;
  return ICS_OK;
}
// the whole u frame (the reference updates the caller's `u` view in place, border ring included) -> dst[y:y+uM, x:x+uN]
extern "C" int ics_rl_download_img(ics_rl* j, ics_img* dst, int y, int x) {
  if (!j || !dst) return fail(ICS_EINVAL, "NULL argument");
  const IcsGeom& g = j->g;
  if (dst->ctx != j->ctx) return fail(ICS_EINVAL, "image of another context");
  if (!rect_ok(dst, y, x, g.uM, g.uN)) return fail(ICS_EINVAL, "u window [%d:%d, %d:%d] outside a %d x %d image", y, y + g.uM, x, x + g.uN, dst->H, dst->W);
  HIPCHK(hipSetDevice(j->ctx->device));
  HIPCHK(hipMemcpy2DAsync(dst->d + ((size_t)y * dst->W + x) * 3, (size_t)dst->W * 12, org(j, j->u), (size_t)g.pitch * 4, (size_t)g.uN * 12, g.uM,
                          hipMemcpyDeviceToDevice, j->ctx->stream));
  return ICS_OK;
}
