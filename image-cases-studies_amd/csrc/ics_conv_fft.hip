// ics_conv_fft.hip -- the PSF convolutions of one Richardson-Lucy inner iteration as LDS-resident overlap-save FFT tiles, gfx950.
//
//   mode 0 (A1+A2 / A11, lib/deconvolution.pyx:477-488, 555-565):  error = convolve(u, psf, "valid") - image
//   mode 1 (A3, pyx:490-491):  gradu = convolve(error, rot180(psf), "full")  (+ the reductions of A7, pyx:523-524)
//
// The reference computes both with scipy's complex64 FFT over the whole frame (pyx:478,491 -> scipy.signal.fftconvolve); here the frame is
// cut into tiles of V = 128 - K + 1 output pixels a side, each the valid part of a 128 x 128 circular correlation (overlap-save), fp32
// throughout.  The matrix-core kernels (ics_conv_mfma.hip) pay 3 split terms x 47..65 % Toeplitz fill -- at 31 x 31 a fifth of their
// MFMA flops is useful and the pass takes 0.9-1.0 ms at 6144^2; a 128 x 128 transform pair costs ~130 flop per output value whatever K is.
//
// In u-frame coordinates (ics_common.h) both modes are  out[y, x, c] = sum_{a,b<K} W[a, b, c] in[y + a - pad, x + b - pad, c]
// (W = rot180(psf) for mode 0, psf for mode 1).  With t = the 128 x 128 window of `in` that starts at (oy - pad, ox - pad),
//     out[oy + v, ox + h] = r[v][h],   r = IDFT( conj(DFT(W)) . DFT(t) ),   valid for v, h < V   (no wrap-around reaches them),
// and S = conj(DFT2(W zero-padded)) / 128^2 is built once per PSF by k_fft_spectrum.
//
// Work unit = (a PAIR of horizontally adjacent tiles, one channel): the two real tiles travel as real and imaginary part of one complex
// tile -- W is real, so IDFT(S . DFT(a + i b)) = corr(a) + i corr(b) with no separation step.  One 1024-thread workgroup per CU holds the
// complex tile in LDS (128 rows x 136 complex = 136 KB; pitch 272 dwords = 16 banks mod 64) and walks units n = r * grid + q, q chosen so that
// the three channel units of a tile pair run at the same time on three CUs of ONE XCD: the HWC lines a channel unit touches (4 of every
// 12 bytes) are the lines its two siblings touch, and they meet in that XCD's L2.
//
// 128 = 16 x 8 per dimension: n = j + 8 m, k = k1 + 16 k2,
//     X[k1 + 16 k2] = sum_j w8^(j k2) [ w128^(j k1) sum_m x[j + 8 m] w16^(m k1) ]            (forward; the inverse runs the same steps backwards)
// so a thread always holds 16 complex values: one radix-16 or two radix-8 transforms, and every exchange goes through LDS:
//   A  x-major (wave: j = w & 7, 64 columns)   global -> radix-16 over m -> twiddle -> LDS row 16 j + k1          | barrier
//   B  x-major (k1 = (w & 7) + 8 s)            radix-8 over j  -> row k1 + 16 k2 (= ky)                            | barrier
//   C  row-owner (wave: 8 rows; j = lane & 7)  radix-16 over m (x = j + 8 m) -> twiddle -> column 8 k1 + (j + k1) % 8
//   D  row-owner (k1 = (lane & 7) + 8 s)       radix-8 over j -> kx = k1 + 16 k2; x S[ky][kx]; inverse radix-8 over k2 -> same slots
//   E  row-owner                               conj twiddle, inverse radix-16 over k1 -> x = j + 8 m              | barrier
//   F  x-major                                 inverse radix-8 over k2 (rows k1 + 16 k2) -> row 16 j + k1         | barrier
//   G  x-major                                 conj twiddle, inverse radix-16 -> y = j + 8 m; epilogue straight from the registers
// C, D, E exchange data inside a wave's own 8 rows only (a wave's LDS operations execute in order): four workgroup barriers per unit.
// The column skew (j + k1) % 8 and the 16-bank pitch make every ds_read_b64 / ds_write_b64 of C, D, E conflict-free.
//
// Epilogues: the arithmetic of ics_conv.hip (mode 0: minus image on the M x N interior; mode 1: raw sums stored, maxima of
// |lambd g + (u - ut)/2| and u per channel; PAM kinds store G = T + lambd g).  Not bit-identical to the direct-sum kernels (an FFT
// rounds differently): held to the same float64 stage gates (tests/test_gpu_stages.py) and run-level goldens.
#include "ics_common.h"
#include "ics_kernels.h"
#include "ics_tw128.h"

#define ICS_FFT_P 128
#define ICS_FFT_PITCH 136
#define ICS_FFT_LDS_BYTES (ICS_FFT_P * ICS_FFT_PITCH * 8 + 128 * 8)   /* the tile + the twiddle table */
#define ICS_FFT_THREADS 1024

typedef float v2f __attribute__((ext_vector_type(2)));

#ifndef ICS_FFT_HD
#define ICS_FFT_HD __host__ __device__ __forceinline__
#endif

#if defined(__HIP_DEVICE_COMPILE__)
#define ICS_FFT_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)
#define ICS_FSUB(a, b) __fsub_rn(a, b)
#define ICS_FADD(a, b) __fadd_rn(a, b)
#define ICS_FMUL(a, b) __fmul_rn(a, b)
#else   /* host pass: the CPU emulation of tools/bench_conv_fft.hip (-ffp-contract=off: the same single roundings) */
#define ICS_FFT_UNIFORM(x) (x)
#define ICS_FSUB(a, b) ((a) - (b))
#define ICS_FADD(a, b) ((a) + (b))
#define ICS_FMUL(a, b) ((a) * (b))
#endif

namespace icsfft {

// Global memory goes through buffer addressing on the device (SGPR resource + 32-bit lane offset): with flat 64-bit pointers the compiler
// keeps one 64-bit VGPR address per access alive across the unit loop and spills them.  Indices count floats from the START of the frame
// buffer (origin offset added: the apron in front of the origin has negative coordinates).  The host pass (CPU emulation) indexes pointers.
#if defined(__HIP_DEVICE_COMPILE__)
typedef __amdgpu_buffer_rsrc_t gbuf;
__device__ __forceinline__ gbuf make_gbuf(const void* p) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7FFFFFFF, 0x00020000); }
__device__ __forceinline__ float ld_f32(gbuf b, int i) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(b, 4 * i, 0, 0)); }
__device__ __forceinline__ void st_f32(gbuf b, int i, float v) { __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), b, 4 * i, 0, 0); }
__device__ __forceinline__ v2f ld_v2f(gbuf b, int i) { return __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(b, 8 * i, 0, 0)); }
#else
typedef const void* gbuf;
static inline gbuf make_gbuf(const void* p) { return p; }
static inline float ld_f32(gbuf b, int i) { return static_cast<const float*>(b)[i]; }
static inline void st_f32(gbuf b, int i, float v) { const_cast<float*>(static_cast<const float*>(b))[i] = v; }
static inline v2f ld_v2f(gbuf b, int i) { return static_cast<const v2f*>(b)[i]; }
#endif
struct Mem {
  gbuf in, out, f, u, ut, tv, spec;
  int org;   // floats from the start of a frame buffer to its origin
};

// exp(-2 pi i t / 128): device copy (scalar / vector loads through the caches) and host copy (CPU emulation in tools/bench_conv_fft.hip)
__device__ __constant__ const float d_tw128[128][2] = {ICS_TW128_VALUES};
static const float h_tw128[128][2] = {ICS_TW128_VALUES};

ICS_FFT_HD v2f tw128(int t) {
#if defined(__HIP_DEVICE_COMPILE__)
  return (v2f){d_tw128[t & 127][0], d_tw128[t & 127][1]};
#else
  return (v2f){h_tw128[t & 127][0], h_tw128[t & 127][1]};
#endif
}

// a * b and a * conj(b): one packed multiply + one packed fma
ICS_FFT_HD v2f cmul(v2f a, v2f b) { return __builtin_elementwise_fma((v2f){a.y, a.y}, (v2f){-b.y, b.x}, (v2f){a.x, a.x} * b); }
ICS_FFT_HD v2f cmulc(v2f a, v2f b) { return __builtin_elementwise_fma((v2f){a.y, a.y}, (v2f){b.y, b.x}, (v2f){a.x, a.x} * (v2f){b.x, -b.y}); }
// forward twiddles are exp(-i phi): DIR = +1 multiplies by b, DIR = -1 by conj(b)
template <int DIR> ICS_FFT_HD v2f cmuld(v2f a, v2f b) { return DIR > 0 ? cmul(a, b) : cmulc(a, b); }
// a * (-i) forward, a * (+i) inverse
template <int DIR> ICS_FFT_HD v2f rot90(v2f a) { return DIR > 0 ? (v2f){a.y, -a.x} : (v2f){-a.y, a.x}; }

template <int DIR> ICS_FFT_HD void fft4(v2f& a0, v2f& a1, v2f& a2, v2f& a3) {
  const v2f t0 = a0 + a2, t1 = a0 - a2, t2 = a1 + a3, t3 = rot90<DIR>(a1 - a3);
  a0 = t0 + t2; a2 = t0 - t2; a1 = t1 + t3; a3 = t1 - t3;
}

// 8 points, natural order in, natural order out.  n = 2 n1 + n2, k = k1 + 4 k2.
template <int DIR> ICS_FFT_HD void fft8(v2f (&v)[8]) {
  constexpr float R = 0.70710678118654752440f;
  v2f e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6], o0 = v[1], o1 = v[3], o2 = v[5], o3 = v[7];
  fft4<DIR>(e0, e1, e2, e3);
  fft4<DIR>(o0, o1, o2, o3);
  // o[k1] *= w8^(k1):  w8 = (1 - i)/sqrt2 forward, (1 + i)/sqrt2 inverse
  if (DIR > 0) {
    o1 = (v2f){o1.x + o1.y, o1.y - o1.x} * R;
    o2 = (v2f){o2.y, -o2.x};
    o3 = (v2f){o3.y - o3.x, -o3.x - o3.y} * R;
  } else {
    o1 = (v2f){o1.x - o1.y, o1.x + o1.y} * R;
    o2 = (v2f){-o2.y, o2.x};
    o3 = (v2f){-o3.x - o3.y, o3.x - o3.y} * R;
  }
  v[0] = e0 + o0; v[4] = e0 - o0;
  v[1] = e1 + o1; v[5] = e1 - o1;
  v[2] = e2 + o2; v[6] = e2 - o2;
  v[3] = e3 + o3; v[7] = e3 - o3;
}

// 16 points, natural order in, natural order out.  n = 4 n1 + n2, k = k1 + 4 k2.
template <int DIR> ICS_FFT_HD void fft16(v2f (&v)[16]) {
  constexpr float C1 = 0.92387953251128675613f, S1 = 0.38268343236508977173f, R = 0.70710678118654752440f;
  v2f a[4][4];   // a[n2][k1]
#pragma unroll
  for (int n2 = 0; n2 < 4; ++n2) {
    a[n2][0] = v[n2]; a[n2][1] = v[4 + n2]; a[n2][2] = v[8 + n2]; a[n2][3] = v[12 + n2];
    fft4<DIR>(a[n2][0], a[n2][1], a[n2][2], a[n2][3]);
  }
  // a[n2][k1] *= w16^(n2 k1), w16^t = (cos(pi t / 8), -sin(pi t / 8)) forward
  a[1][1] = cmuld<DIR>(a[1][1], (v2f){C1, -S1});
  a[1][2] = cmuld<DIR>(a[1][2], (v2f){R, -R});
  a[1][3] = cmuld<DIR>(a[1][3], (v2f){S1, -C1});
  a[2][1] = cmuld<DIR>(a[2][1], (v2f){R, -R});
  a[2][2] = rot90<DIR>(a[2][2]);
  a[2][3] = cmuld<DIR>(a[2][3], (v2f){-R, -R});
  a[3][1] = cmuld<DIR>(a[3][1], (v2f){S1, -C1});
  a[3][2] = cmuld<DIR>(a[3][2], (v2f){-R, -R});
  a[3][3] = cmuld<DIR>(a[3][3], (v2f){-C1, S1});
#pragma unroll
  for (int k1 = 0; k1 < 4; ++k1) {
    fft4<DIR>(a[0][k1], a[1][k1], a[2][k1], a[3][k1]);   // -> k2 = 0..3
#pragma unroll
    for (int k2 = 0; k2 < 4; ++k2) v[k1 + 4 * k2] = a[k2][k1];
  }
}

}  // namespace icsfft

// ---- arguments ---------------------------------------------------------------------------------------------------------------------------
struct IcsFftArgs {
  IcsConvArgs c;        // frames, operands, reduction slots, geometry (c.w / c.bt / c.facc / c.sched unused)
  const v2f* spec;      // [3][128][128]: conj(DFT2(W_c)) / 128^2 of this orientation (k_fft_spectrum)
  int V;                // valid output pixels per tile edge = 128 - K + 1
  int tiles_x, ntiles, nunits;
  int oy0, ox0, oy1, ox1;   // output region in u-frame coordinates (mode 0: the M x N interior; mode 1: the whole u-frame)
};

namespace icsfft {

struct Unit {
  int c;            // channel
  int oy[2], ox[2]; // u-frame coordinates of output pixel (0, 0) of the two tiles
  bool has[2];
};

ICS_FFT_HD Unit decode_unit(const IcsFftArgs& a, int n) {
  Unit u;
  const int pair = n / 3;
  u.c = n - 3 * pair;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int ti = 2 * pair + t;
    u.has[t] = ti < a.ntiles;
    const int ty = ti / a.tiles_x, tx = ti - ty * a.tiles_x;
    u.oy[t] = a.oy0 + ty * a.V; u.ox[t] = a.ox0 + tx * a.V;
  }
  return u;
}

ICS_FFT_HD Mem make_mem(const IcsFftArgs& a) {
  Mem m;
  m.org = a.c.g.ay * a.c.g.pitch + 3 * a.c.g.ax;   // ics_origin_offset()
  m.in = make_gbuf(a.c.in - m.org); m.out = make_gbuf(a.c.out - m.org); m.f = make_gbuf(a.c.f - m.org);
  m.u = make_gbuf(a.c.u - m.org); m.ut = make_gbuf(a.c.ut - m.org); m.tv = make_gbuf(a.c.tv ? a.c.tv - m.org : a.c.u - m.org);
  m.spec = make_gbuf(a.spec);
  return m;
}

// x-major mapping (stages A, B, F, G): wave w -> selector w & 7 and columns 64 (w >> 3) + lane
// row-owner mapping (stages C, D, E): wave w -> rows 8 w + (lane >> 3), selector lane & 7
#define ICS_FFT_AT(row, col) lds[(row) * ICS_FFT_PITCH + (col)]

// A: the window's column x, rows j + 8 m of both tiles -> radix-16 over m -> twiddle -> rows 16 j + k1
// (loads from clamped addresses, values selected: no divergent or conditional loads; offsets are 32-bit -- frames stay below 2 GiB)
ICS_FFT_HD void stage_a(const IcsFftArgs& a, const Mem& mem, const Unit& u, v2f* lds, int tid) {
  const int w = ICS_FFT_UNIFORM(tid >> 6), lane = tid & 63, j = w & 7, x = 64 * (w >> 3) + lane;
  const int pad = a.c.g.pad, pitch = a.c.g.pitch, ylast = a.c.g.uM + pad - 1;
  v2f v[16];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int X = u.ox[t] - pad + x;
    const bool xin = u.has[t] && X < a.c.g.uN + pad;         // (X >= -pad by construction)
    const int xo = mem.org + 3 * (xin ? X : 0) + u.c;
    const int Y0 = u.oy[t] - pad + j;                        // >= -pad by construction
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      const int Y = Y0 + 8 * m;
      const float val = ld_f32(mem.in, (Y < ylast ? Y : ylast) * pitch + xo);
      const float sel = (xin && Y <= ylast) ? val : 0.f;
      if (t == 0) v[m].x = sel; else v[m].y = sel;
    }
  }
  fft16<1>(v);
#pragma unroll
  for (int k1 = 0; k1 < 16; ++k1) {
    const v2f r = k1 ? cmul(v[k1], tw128(j * k1)) : v[k1];
    ICS_FFT_AT(16 * j + k1, x) = r;
  }
}

// B: radix-8 over j at fixed k1 -> rows ky = k1 + 16 k2  (the same eight slots)
template <int DIR> ICS_FFT_HD void stage_b(v2f* lds, int tid) {
  const int w = ICS_FFT_UNIFORM(tid >> 6), lane = tid & 63, x = 64 * (w >> 3) + lane;
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int k1 = (w & 7) + 8 * s;
    v2f v[8];
    if (DIR > 0) {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = ICS_FFT_AT(16 * j + k1, x);
      fft8<1>(v);
#pragma unroll
      for (int k2 = 0; k2 < 8; ++k2) ICS_FFT_AT(k1 + 16 * k2, x) = v[k2];
    } else {   // F: inverse radix-8 over k2 -> rows 16 j + k1
#pragma unroll
      for (int k2 = 0; k2 < 8; ++k2) v[k2] = ICS_FFT_AT(k1 + 16 * k2, x);
      fft8<-1>(v);
#pragma unroll
      for (int j = 0; j < 8; ++j) ICS_FFT_AT(16 * j + k1, x) = v[j];
    }
  }
}

ICS_FFT_HD int skew_col(int j, int k1) { return 8 * k1 + ((j + k1) & 7); }

// C: row ky, x = j + 8 m -> radix-16 over m -> twiddle -> column 8 k1 + (j + k1) % 8
// (`rd` = `lds` on the device -- the lanes of a wave run in lock step, every read is back before the first write; the CPU emulation, which
//  runs the threads one after the other, passes a snapshot)
// (`twl` = the 128 twiddles in LDS behind the tile: the lane-dependent ones of C and E are read from there)
ICS_FFT_HD void stage_c(const v2f* rd, v2f* lds, const v2f* twl, int tid) {
  const int w = ICS_FFT_UNIFORM(tid >> 6), lane = tid & 63, ky = 8 * w + (lane >> 3), j = lane & 7;
  v2f v[16];
#pragma unroll
  for (int m = 0; m < 16; ++m) v[m] = rd[ky * ICS_FFT_PITCH + j + 8 * m];
  fft16<1>(v);
#pragma unroll
  for (int k1 = 0; k1 < 16; ++k1) ICS_FFT_AT(ky, skew_col(j, k1)) = k1 ? cmul(v[k1], twl[(j * k1) & 127]) : v[k1];
}

// D: radix-8 over j -> kx = k1 + 16 k2, multiply by the spectrum, inverse radix-8 over k2 -> j, same slots
ICS_FFT_HD void stage_d(const Mem& mem, int c, v2f* lds, int tid) {
  const int w = ICS_FFT_UNIFORM(tid >> 6), lane = tid & 63, ky = 8 * w + (lane >> 3), q = lane & 7;
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int k1 = q + 8 * s;
    v2f v[8], sp[8];
#pragma unroll
    for (int k2 = 0; k2 < 8; ++k2) sp[k2] = ld_v2f(mem.spec, (c * ICS_FFT_P + ky) * ICS_FFT_P + k1 + 16 * k2);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = ICS_FFT_AT(ky, skew_col(j, k1));
    fft8<1>(v);
#pragma unroll
    for (int k2 = 0; k2 < 8; ++k2) v[k2] = cmul(v[k2], sp[k2]);
    fft8<-1>(v);
#pragma unroll
    for (int j = 0; j < 8; ++j) ICS_FFT_AT(ky, skew_col(j, k1)) = v[j];
  }
}

// E: conj twiddle, inverse radix-16 over k1 -> x = j + 8 m
ICS_FFT_HD void stage_e(const v2f* rd, v2f* lds, const v2f* twl, int tid) {
  const int w = ICS_FFT_UNIFORM(tid >> 6), lane = tid & 63, ky = 8 * w + (lane >> 3), j = lane & 7;
  v2f v[16];
#pragma unroll
  for (int k1 = 0; k1 < 16; ++k1) {
    const v2f r = rd[ky * ICS_FFT_PITCH + skew_col(j, k1)];
    v[k1] = k1 ? cmulc(r, twl[(j * k1) & 127]) : r;
  }
  fft16<-1>(v);
#pragma unroll
  for (int m = 0; m < 16; ++m) ICS_FFT_AT(ky, j + 8 * m) = v[m];
}

// G, first half: conj twiddle, inverse radix-16 over k1 -> rows y = j + 8 m of column x (tile 0 in .x, tile 1 in .y)
ICS_FFT_HD void stage_g(const v2f* lds, int tid, v2f (&v)[16]) {
  const int w = ICS_FFT_UNIFORM(tid >> 6), lane = tid & 63, j = w & 7, x = 64 * (w >> 3) + lane;
#pragma unroll
  for (int k1 = 0; k1 < 16; ++k1) {
    const v2f r = ICS_FFT_AT(16 * j + k1, x);
    v[k1] = k1 ? cmulc(r, tw128(j * k1)) : r;
  }
  fft16<-1>(v);
}

// canonical positive NaN so that a NaN propagates through the integer max like np.amax does (ics_conv.hip)
ICS_FFT_HD uint32_t key_of(float f) { return (f != f) ? 0xFFC00000u : ics_f2key(f); }

// G, second half: the epilogue of ics_conv.hip on the thread's 16 + 16 values (rows j + 8 m of column x; j is wave-uniform, so the
// row count is a scalar).  Accumulates the thread's maxima (mode 1).
template <int MODE>
ICS_FFT_HD void epilogue(const IcsFftArgs& a, const Mem& mem, const Unit& u, int tid, const v2f (&v)[16], float& mg, float& mu, bool& nan_g, bool& nan_u, bool& any) {
  const int w = ICS_FFT_UNIFORM(tid >> 6), lane = tid & 63, j = w & 7, x = 64 * (w >> 3) + lane;
  const int pitch = a.c.g.pitch, c = u.c, dstep = 8 * pitch;
  const float lambd = a.c.lambd;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int X = u.ox[t] + x;
    const bool xok = u.has[t] && x < a.V && X < a.ox1;
    const int rows = a.oy1 - u.oy[t] < a.V ? a.oy1 - u.oy[t] : a.V;   // output rows of this tile
    const int mcount = (rows - j + 7) >> 3;                            // of them, this wave's: y = j + 8 m < rows
    if (!xok || mcount <= 0) continue;
    const int o0 = mem.org + (u.oy[t] + j) * pitch + 3 * X + c;
    if (MODE == 0) {
      float f[16];
#pragma unroll
      for (int m = 0; m < 16; ++m) if (m < mcount) f[m] = ld_f32(mem.f, o0 + m * dstep);
#pragma unroll
      for (int m = 0; m < 16; ++m) if (m < mcount) st_f32(mem.out, o0 + m * dstep, ICS_FSUB(t ? v[m].y : v[m].x, f[m]));
    } else {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        float uu[8], tt[8], tv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int m = 8 * h + i;
          if (m < mcount) { uu[i] = ld_f32(mem.u, o0 + m * dstep); tt[i] = ld_f32(mem.ut, o0 + m * dstep); tv[i] = a.c.tv_kind ? ld_f32(mem.tv, o0 + m * dstep) : 0.f; }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int m = 8 * h + i;
          if (m < mcount) {
            const int Y = u.oy[t] + j + 8 * m;
            const float r = t ? v[m].y : v[m].x;
            float g, st = r;
            if (a.c.tv_kind >= 2) { g = (float)((double)tv[i] + (double)ICS_FMUL(lambd, r)); st = g; }          // PAM: G = T + lambd*gradu, stored
            else if (a.c.tv_kind == 1 && Y >= 1 && Y <= a.c.g.uM - 2 && X >= 1 && X <= a.c.g.uN - 2)             // active MM-TV, pyx:517
              g = (float)(((double)tv[i] + (double)ICS_FMUL(lambd, r)) + (double)ICS_FSUB(uu[i], tt[i]) / 4.0);
            else
              g = ICS_FADD(ICS_FMUL(lambd, r), ICS_FMUL(ICS_FSUB(uu[i], tt[i]), 0.5f));                           // pyx:519
            mg = __builtin_fmaxf(mg, __builtin_fabsf(g));
            mu = __builtin_fmaxf(mu, uu[i]);
            nan_g |= (g != g); nan_u |= (uu[i] != uu[i]);
            any = true;
            st_f32(mem.out, o0 + m * dstep, st);
          }
        }
      }
    }
  }
}

}  // namespace icsfft

#if defined(__HIPCC__)
namespace icsfft {

__device__ __forceinline__ void wave_sync() {
  // stages C, D, E exchange data between the lanes of ONE wave through LDS: a wave's DS operations execute in order, the compiler must
  // keep them in program order
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// workgroup barrier that waits for this wave's LDS traffic only (__syncthreads() also waits for the global loads and stores in flight)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int MODE>
__global__ __launch_bounds__(ICS_FFT_THREADS) void k_conv_fft(IcsFftArgs a) {
  extern __shared__ __attribute__((aligned(16))) v2f lds[];
  v2f* const twl = lds + ICS_FFT_P * ICS_FFT_PITCH;
  const int tid = threadIdx.x;
  const int G = gridDim.x;
  if (tid < 128) twl[tid] = tw128(tid);
  const Mem mem = make_mem(a);
  // workgroup b runs on XCD b % 8 (observed dispatch): consecutive unit slots q go to one XCD, so the three channel units of a tile pair
  // (n = 3 pair + c) share that XCD's L2.  Affects speed only.
  const int q = (G & 7) ? (int)blockIdx.x : (int)(blockIdx.x & 7) * (G >> 3) + (int)(blockIdx.x >> 3);
  for (int n = q; n < a.nunits; n += G) {
    const Unit u = decode_unit(a, n);
    stage_a(a, mem, u, lds, tid);
    lds_barrier();
    stage_b<1>(lds, tid);
    lds_barrier();
    stage_c(lds, lds, twl, tid);
    wave_sync();
    stage_d(mem, u.c, lds, tid);
    wave_sync();
    stage_e(lds, lds, twl, tid);
    lds_barrier();
    stage_b<-1>(lds, tid);
    lds_barrier();
    v2f v[16];
    stage_g(lds, tid, v);
    float mg = 0.f, mu = -__builtin_inff();
    bool nan_g = false, nan_u = false, any = false;
    epilogue<MODE>(a, mem, u, tid, v, mg, mu, nan_g, nan_u, any);
    if (MODE == 1) {
      // wave maxima -> one conditional atomic per wave and value (the running maximum only grows: a stale read lets most skip the atomic)
      uint32_t kg = nan_g ? 0xFFC00000u : (any ? ics_f2key(mg) : 0u);
      uint32_t ku = nan_u ? 0xFFC00000u : (any ? ics_f2key(mu) : 0u);
      kg = ics_wave_max_u32(kg); ku = ics_wave_max_u32(ku);
      if ((tid & 63) == 0) {
        if (kg > a.c.red[ICS_RED_MAXG + u.c]) atomicMax(a.c.red + ICS_RED_MAXG + u.c, kg);
        if (ku > a.c.red[ICS_RED_MAXU + u.c]) atomicMax(a.c.red + ICS_RED_MAXU + u.c, ku);
      }
    }
  }
}

// ---- spectrum: S[o][c][ky][kx] = conj( sum_{a,b} W_o[a][b][c] w^(a ky + b kx) ) / 128^2,  w = exp(-2 pi i / 128) ------------------------------
// W_0 = rot180(psf) (mode 0), W_1 = psf (mode 1).  Double accumulation (a PSF value enters with its float32 value, the twiddles from a
// double table built on the device); one workgroup per (orientation, channel, 32 columns kx): G[a][kx] = sum_b W[a][b] w^(b kx) in LDS,
// then S[ky][kx] = conj(sum_a G[a][kx] w^(a ky)).
__global__ __launch_bounds__(256) void k_fft_spectrum(const float* __restrict__ psf, int K, v2f* __restrict__ spec0, v2f* __restrict__ spec1) {
  extern __shared__ __attribute__((aligned(16))) double sm[];   // [128][2] twiddles, then [K][32][2] G
  double* twd = sm;
  double* Gs = sm + 256;
  const int tid = threadIdx.x;
  const int o = blockIdx.x / 12, c = (blockIdx.x / 4) % 3, kx0 = (blockIdx.x & 3) * 32;
  if (tid < 128) {
    double sn, cs;
    sincospi((double)tid / 64.0, &sn, &cs);
    twd[2 * tid] = cs; twd[2 * tid + 1] = -sn;
  }
  __syncthreads();
  for (int i = tid; i < K * 32; i += 256) {
    const int aa = i >> 5, kx = kx0 + (i & 31);
    double re = 0.0, im = 0.0;
    for (int b = 0; b < K; ++b) {
      const double wv = o == 0 ? (double)psf[((K - 1 - aa) * K + (K - 1 - b)) * 3 + c] : (double)psf[(aa * K + b) * 3 + c];
      const int t = (b * kx) & 127;
      re += wv * twd[2 * t]; im += wv * twd[2 * t + 1];
    }
    Gs[2 * i] = re; Gs[2 * i + 1] = im;
  }
  __syncthreads();
  v2f* out = (o == 0 ? spec0 : spec1) + (size_t)c * ICS_FFT_P * ICS_FFT_P;
  for (int i = tid; i < 128 * 32; i += 256) {
    const int ky = i >> 5, kxl = i & 31;
    double re = 0.0, im = 0.0;
    for (int aa = 0; aa < K; ++aa) {
      const double gr = Gs[2 * (aa * 32 + kxl)], gi = Gs[2 * (aa * 32 + kxl) + 1];
      const int t = (aa * ky) & 127;
      const double wr = twd[2 * t], wi = twd[2 * t + 1];
      re += gr * wr - gi * wi; im += gr * wi + gi * wr;
    }
    const double sc = 1.0 / (128.0 * 128.0);
    out[ky * ICS_FFT_P + kx0 + kxl] = (v2f){(float)(re * sc), (float)(-im * sc)};
  }
}

}  // namespace icsfft

// ---- launchers -----------------------------------------------------------------------------------------------------------------------------
bool ics_conv_fft_supported(int K) { return K >= 3 && K <= 65 && (K & 1); }
size_t ics_conv_fft_spectrum_floats() { return (size_t)3 * ICS_FFT_P * ICS_FFT_P * 2; }   // per orientation

hipError_t ics_launch_fft_spectrum(const float* psf, int K, float* spec_conv, float* spec_corr, hipStream_t s) {
  const size_t lds = (256 + (size_t)K * 32 * 2) * sizeof(double);   // 35 KB at K = 65
  hipLaunchKernelGGL(icsfft::k_fft_spectrum, dim3(24), dim3(256), lds, s, psf, K, reinterpret_cast<v2f*>(spec_conv), reinterpret_cast<v2f*>(spec_corr));
  return hipGetLastError();
}

void ics_conv_fft_fill_args(int mode, const IcsConvArgs& c, const float* spec, IcsFftArgs* a) {
  a->c = c;
  a->spec = reinterpret_cast<const v2f*>(spec);
  const IcsGeom& g = c.g;
  a->V = ICS_FFT_P - g.K + 1;
  if (mode == 0) { a->oy0 = g.pad; a->ox0 = g.pad; a->oy1 = g.pad + g.M; a->ox1 = g.pad + g.N; }
  else { a->oy0 = 0; a->ox0 = 0; a->oy1 = g.uM; a->ox1 = g.uN; }
  a->tiles_x = (a->ox1 - a->ox0 + a->V - 1) / a->V;
  const int tiles_y = (a->oy1 - a->oy0 + a->V - 1) / a->V;
  a->ntiles = a->tiles_x * tiles_y;
  a->nunits = 3 * ((a->ntiles + 1) / 2);
}

hipError_t ics_launch_conv_fft(int mode, const IcsConvArgs& c, const float* spec, hipStream_t s) {
  if (mode != 0 && mode != 1) return hipErrorInvalidValue;
  IcsFftArgs a;
  ics_conv_fft_fill_args(mode, c, spec, &a);
  static std::atomic<bool> configured[2][ICS_MAX_DEVICES];
  const int dev = ics_current_device();
  int grid = ics_device_cus(dev);
  if (const int mw = ics_debug().max_wgs.load(std::memory_order_relaxed); mw > 0 && grid > mw) grid = mw;
  if (grid > a.nunits) grid = a.nunits;
  if (mode == 0) {
    if (hipError_t e = ics_configure_lds(configured[0], dev, icsfft::k_conv_fft<0>, ICS_FFT_LDS_BYTES); e != hipSuccess) return e;
    hipLaunchKernelGGL(icsfft::k_conv_fft<0>, dim3(grid), dim3(ICS_FFT_THREADS), ICS_FFT_LDS_BYTES, s, a);
  } else {
    if (hipError_t e = ics_configure_lds(configured[1], dev, icsfft::k_conv_fft<1>, ICS_FFT_LDS_BYTES); e != hipSuccess) return e;
    hipLaunchKernelGGL(icsfft::k_conv_fft<1>, dim3(grid), dim3(ICS_FFT_THREADS), ICS_FFT_LDS_BYTES, s, a);
  }
  return hipGetLastError();
}
#endif
