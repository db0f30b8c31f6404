// ics_conv_fft.hip -- the PSF convolutions of one Richardson-Lucy inner iteration as LDS-resident overlap-save FFT tiles, gfx950.
//
//   mode 0 (A1+A2 / A11, lib/deconvolution.pyx:477-488, 555-565):  error = convolve(u, psf, "valid") - image
//   mode 1 (A3, pyx:490-491):  gradu = convolve(error, rot180(psf), "full")  (+ the reductions of A7, pyx:523-524)
//   mode 2 (A1+A2+A3 in one unit, round 6):  gradu straight from u and the image -- interior tiles never leave the frequency domain,
//                                             G = S1 (16384 S0 T - F) with F = the image windows' spectra (k_fft_image_spectrum); see tile_is_border
//   k_synth_gradk_fft (A11+A12+A13 in one unit, round 6), k_gradk_fft (A12+A13), k_fft_spectrum (the weight spectra): further down
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
#include <algorithm>
#include <vector>

#define ICS_FFT_P 128
#ifndef ICS_FFT_MAX_K
#define ICS_FFT_MAX_K 85        /* largest PSF size ONE tile takes (44 valid pixels a side); above it the PSF is cut into tap blocks (k_conv_fft_blk), which measured
                                   ahead from about there: 4096^2 non-blind 85 one tile 1.68 ms, 97 one tile 2.85, 99 as 2 x 2 blocks 1.59 */
#endif
#define ICS_FFT_PITCH 136
#define ICS_FFT_TWS 17         /* the twiddle table behind the tile: T[j][k1] = w^(j k1), j < 8, k1 < 16, rows of 17 entries (34 dwords: the eight j of a
                                  wave's lanes fall into different banks), so that a lane's fifteen reads are ONE address + immediate offsets */
#define ICS_FFT_TW_ENTRIES (8 * ICS_FFT_TWS)
#define ICS_FFT_LDS_BYTES (ICS_FFT_P * ICS_FFT_PITCH * 8 + ICS_FFT_TW_ENTRIES * 8)   /* the tile + the twiddle table */
#define ICS_FFT_THREADS 1024

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

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

// Global memory goes through buffer addressing on the device (SGPR resource + 32-bit lane offset + SGPR offset): with flat 64-bit pointers
// the compiler keeps one 64-bit VGPR address per access alive across the unit loop and spills them, and every access costs vector
// instructions for its address.  Here an access is  base + 4 * (lane index) + 4 * (wave-uniform index)  with the uniform part in a scalar
// register: the row walk of a tile costs no vector instruction at all.  Indices count floats from the START of the frame buffer (origin
// offset added: the apron in front of the origin has negative coordinates).
// Lane index ICS_FFT_NONE = "no access": its byte offset 2^31 lies beyond num_records, the hardware returns 0 for the load and drops the
// store.  Every access is issued unconditionally, so the number of memory operations in flight is static and the compiler's
// s_waitcnt vmcnt(n) for the register prefetch of the next unit does not degrade to vmcnt(0) behind the epilogue's stores.
// The host pass (CPU emulation, tools/bench_conv_fft.hip) indexes pointers.
#define ICS_FFT_NONE 0x20000000
// Measurement hooks (per-wave phase timeline, ablations): empty in the library; the harness builds of tools/bench_conv_fft.hip define
// ICS_FFT_PROBES and get their bodies from tools/ics_conv_fft_probe.h.
#ifdef ICS_FFT_PROBES
#include "tools/ics_conv_fft_probe.h"
#else
#define ICS_FFT_PROBE_SKIP_LOAD(KIND, vi, si)
#define ICS_FFT_PROBE_SKIP_STORE(v, vi, si)
#define ICS_FFT_PROBE_SKIP_MATH() do { } while (0)
#define ICS_FFT_PROBE_COLUMN_PASS(x) do { x } while (0)
#define ICS_FFT_PROBE_TRACE_DECL() do { } while (0)
#define ICS_FFT_STAMP(i) do { } while (0)
#define ICS_FFT_PROBE_TRACE_NEXT() do { } while (0)
#define ICS_FFT_PROBE_STAGGER() do { } while (0)
#endif
#if defined(__HIP_DEVICE_COMPILE__)
typedef __amdgpu_buffer_rsrc_t gbuf;
__device__ __forceinline__ gbuf make_gbuf(const void* p) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7FFFFFFF, 0x00020000); }
__device__ __forceinline__ float ld_f32(gbuf b, int vi, int si) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(b, 4 * vi, 4 * si, 0)); }
__device__ __forceinline__ void st_f32(gbuf b, int vi, int si, float v) { __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), b, 4 * vi, 4 * si, 0); }
__device__ __forceinline__ v2f ld_v2f(gbuf b, int vi, int si) { return __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(b, 8 * vi, 8 * si, 0)); }
template <int KIND = 0>   // (KIND: which class of access this is -- spectrum 1, operands 2 / 16, window 4 -- for the harness' ablation hook)
__device__ __forceinline__ v4f ld_f32x4(gbuf b, int vi, int si) {
  ICS_FFT_PROBE_SKIP_LOAD(KIND, vi, si)
  return __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(b, 4 * vi, 4 * si, 0));
}
typedef uint32_t u4v __attribute__((ext_vector_type(4)));
// (the s_nop behind the store, with the data registers as its operands: a buffer store of more than 8 bytes reads its data registers for a
//  few cycles after it issues, and a vector instruction that rewrites one of them right behind it changes what is stored.  The compiler's
//  hazard table inserts wait states for that -- except when the store has an SGPR offset, which it takes to be safe.  On MI355X it is
//  not: `buffer_store_dwordx4 v[50:53], v58, s[20:23], s29 offen` followed by `v_mov_b32 v50, v0` stored the new v50 on some lanes
//  (tools/bench_conv_fft.hip found it: the first pixel of the quads of lanes 12-15 of every row group but the first).  Keeping the data
//  alive across two wait states costs nothing here: eight stores per thread and unit)
__device__ __forceinline__ void st_f32x4(gbuf b, int vi, int si, v4f v) {
  ICS_FFT_PROBE_SKIP_STORE(v, vi, si)
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4v, v), b, 4 * vi, 4 * si, 0);
  asm volatile("s_nop 2" :: "v"(v) : "memory");
}
#else
typedef const void* gbuf;
static inline gbuf make_gbuf(const void* p) { return p; }
static inline float ld_f32(gbuf b, int vi, int si) { return vi >= ICS_FFT_NONE ? 0.f : static_cast<const float*>(b)[vi + si]; }
static inline void st_f32(gbuf b, int vi, int si, float v) { if (vi < ICS_FFT_NONE) const_cast<float*>(static_cast<const float*>(b))[vi + si] = v; }
static inline v2f ld_v2f(gbuf b, int vi, int si) { return static_cast<const v2f*>(b)[vi + si]; }
template <int KIND = 0>
static inline v4f ld_f32x4(gbuf b, int vi, int si) {
  if (vi >= ICS_FFT_NONE) return (v4f){0.f, 0.f, 0.f, 0.f};
  const float* p = static_cast<const float*>(b) + vi + si;
  return (v4f){p[0], p[1], p[2], p[3]};
}
static inline void st_f32x4(gbuf b, int vi, int si, v4f v) {
  if (vi >= ICS_FFT_NONE) return;
  float* p = const_cast<float*>(static_cast<const float*>(b)) + vi + si;
  p[0] = v.x; p[1] = v.y; p[2] = v.z; p[3] = v.w;
}
#endif
// where pixel (Y, X, c) of a channel-planar frame lives: index = org + Y * pitch + X + c * cmul  (ics_common.h: ics_ppitch, ics_plane_floats)
struct Lay { int org, pitch, cmul; };
struct Mem {
  gbuf in, out, f, u, ut, tv, spec, spec1, fspec;
  Lay lin, lout, lf, lu, lut, ltv;
};

// exp(-2 pi i t / 128): device copy (scalar / vector loads through the caches) and host copy (CPU emulation in tools/bench_conv_fft.hip)
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __constant__ const float d_tw128[128][2] = {ICS_TW128_VALUES};
#else
static const float h_tw128[128][2] = {ICS_TW128_VALUES};
#endif

ICS_FFT_HD v2f tw128(int t) {
#if defined(__HIP_DEVICE_COMPILE__)
  return (v2f){d_tw128[t & 127][0], d_tw128[t & 127][1]};
#else
  return (v2f){h_tw128[t & 127][0], h_tw128[t & 127][1]};
#endif
}

// a * b and a * conj(b): one packed multiply + one packed fma.  On the device the operand swaps and sign flips ride on the VOP3P modifiers
// (op_sel / neg): as vector shuffles the compiler spent a v_mov + v_xor on every product with a register operand.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ v2f cmul(v2f a, v2f b) {
  v2f t, r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "v"(b));                                   // (a.x b.x, a.x b.y)
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(a), "v"(b), "v"(t));      // + (-a.y b.y, a.y b.x)
  return r;
}
__device__ __forceinline__ v2f cmulc(v2f a, v2f b) {
  v2f t, r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(t) : "v"(a), "v"(b));                       // (a.x b.x, -a.x b.y)
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(a), "v"(b), "v"(t));                    // + (a.y b.y, a.y b.x)
  return r;
}
// the same with the second factor in a scalar register pair (wave-uniform twiddles of stages A and G: one scalar operand per instruction)
__device__ __forceinline__ v2f cmul_s(v2f a, v2f b) {
  v2f t, r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "s"(b));
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(a), "s"(b), "v"(t));
  return r;
}
__device__ __forceinline__ v2f cmulc_s(v2f a, v2f b) {
  v2f t, r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(t) : "v"(a), "s"(b));
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(a), "s"(b), "v"(t));
  return r;
}
// a + b * (-i) = a + (b.y, -b.x)   and   a + b * (+i) = a + (-b.y, b.x): one instruction each
__device__ __forceinline__ v2f add_mi(v2f a, v2f b) { v2f r; asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ v2f add_pi(v2f a, v2f b) { v2f r; asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b)); return r; }
#else
ICS_FFT_HD v2f cmul(v2f a, v2f b) { return __builtin_elementwise_fma((v2f){a.y, a.y}, (v2f){-b.y, b.x}, (v2f){a.x, a.x} * b); }
ICS_FFT_HD v2f cmulc(v2f a, v2f b) { return __builtin_elementwise_fma((v2f){a.y, a.y}, (v2f){b.y, b.x}, (v2f){a.x, a.x} * (v2f){b.x, -b.y}); }
ICS_FFT_HD v2f cmul_s(v2f a, v2f b) { return cmul(a, b); }
ICS_FFT_HD v2f cmulc_s(v2f a, v2f b) { return cmulc(a, b); }
ICS_FFT_HD v2f add_mi(v2f a, v2f b) { return (v2f){a.x + b.y, a.y - b.x}; }
ICS_FFT_HD v2f add_pi(v2f a, v2f b) { return (v2f){a.x - b.y, a.y + b.x}; }
#endif
// products with COMPILE-TIME constants stay in C++: the compiler folds the swapped / negated constant and reads it from scalar registers
ICS_FFT_HD v2f cmulk(v2f a, v2f b) { return __builtin_elementwise_fma((v2f){a.y, a.y}, (v2f){-b.y, b.x}, (v2f){a.x, a.x} * b); }
ICS_FFT_HD v2f cmulck(v2f a, v2f b) { return __builtin_elementwise_fma((v2f){a.y, a.y}, (v2f){b.y, b.x}, (v2f){a.x, a.x} * (v2f){b.x, -b.y}); }
// forward twiddles are exp(-i phi): DIR = +1 multiplies by b, DIR = -1 by conj(b)
template <int DIR> ICS_FFT_HD v2f cmuld(v2f a, v2f b) { return DIR > 0 ? cmulk(a, b) : cmulck(a, b); }
// a + b * (-i)^DIR and a - b * (-i)^DIR
template <int DIR> ICS_FFT_HD v2f add_rot(v2f a, v2f b) { return DIR > 0 ? add_mi(a, b) : add_pi(a, b); }
template <int DIR> ICS_FFT_HD v2f sub_rot(v2f a, v2f b) { return DIR > 0 ? add_pi(a, b) : add_mi(a, b); }

template <int DIR> ICS_FFT_HD void fft4(v2f& a0, v2f& a1, v2f& a2, v2f& a3) {
  const v2f t0 = a0 + a2, t1 = a0 - a2, t2 = a1 + a3, d = a1 - a3;
  a0 = t0 + t2; a2 = t0 - t2; a1 = add_rot<DIR>(t1, d); a3 = sub_rot<DIR>(t1, d);
}
// the same with input 2 still to be multiplied by (-i)^DIR (w16^4 of the 16-point transform)
template <int DIR> ICS_FFT_HD void fft4_r2(v2f& a0, v2f& a1, v2f& a2, v2f& a3) {
  const v2f t0 = add_rot<DIR>(a0, a2), t1 = sub_rot<DIR>(a0, a2), t2 = a1 + a3, d = a1 - a3;
  a0 = t0 + t2; a2 = t0 - t2; a1 = add_rot<DIR>(t1, d); a3 = sub_rot<DIR>(t1, d);
}

// 8 points, natural order in, natural order out.  n = 2 n1 + n2, k = k1 + 4 k2.
template <int DIR> ICS_FFT_HD void fft8(v2f (&v)[8]) {
  ICS_FFT_PROBE_SKIP_MATH();
  constexpr float R = 0.70710678118654752440f;
  v2f e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6], o0 = v[1], o1 = v[3], o2 = v[5], o3 = v[7];
  fft4<DIR>(e0, e1, e2, e3);
  fft4<DIR>(o0, o1, o2, o3);
  // o[k1] *= w8^(k1):  w8 = (1 - i)/sqrt2 forward, (1 + i)/sqrt2 inverse;  w8^2 = -+i rides on the last butterfly;  w8^3 = -(1 + i)/sqrt2 / -(1 - i)/sqrt2
  o1 = add_rot<DIR>(o1, o1) * R;
  o3 = sub_rot<DIR>(o3, o3) * -R;
  v[0] = e0 + o0; v[4] = e0 - o0;
  v[1] = e1 + o1; v[5] = e1 - o1;
  v[2] = add_rot<DIR>(e2, o2); v[6] = sub_rot<DIR>(e2, o2);
  v[3] = e3 + o3; v[7] = e3 - o3;
}

// 16 points, natural order in, natural order out.  n = 4 n1 + n2, k = k1 + 4 k2.
template <int DIR> ICS_FFT_HD void fft16(v2f (&v)[16]) {
  ICS_FFT_PROBE_SKIP_MATH();
  constexpr float C1 = 0.92387953251128675613f, S1 = 0.38268343236508977173f, R = 0.70710678118654752440f;
  v2f a[4][4];   // a[n2][k1]
#pragma unroll
  for (int n2 = 0; n2 < 4; ++n2) {
    a[n2][0] = v[n2]; a[n2][1] = v[4 + n2]; a[n2][2] = v[8 + n2]; a[n2][3] = v[12 + n2];
    fft4<DIR>(a[n2][0], a[n2][1], a[n2][2], a[n2][3]);
  }
  // a[n2][k1] *= w16^(n2 k1), w16^t = (cos(pi t / 8), -sin(pi t / 8)) forward  (w16^4 = -+i: inside fft4_r2)
  a[1][1] = cmuld<DIR>(a[1][1], (v2f){C1, -S1});
  a[1][2] = cmuld<DIR>(a[1][2], (v2f){R, -R});
  a[1][3] = cmuld<DIR>(a[1][3], (v2f){S1, -C1});
  a[2][1] = cmuld<DIR>(a[2][1], (v2f){R, -R});
  a[2][3] = cmuld<DIR>(a[2][3], (v2f){-R, -R});
  a[3][1] = cmuld<DIR>(a[3][1], (v2f){S1, -C1});
  a[3][2] = cmuld<DIR>(a[3][2], (v2f){-R, -R});
  a[3][3] = cmuld<DIR>(a[3][3], (v2f){-C1, S1});
#pragma unroll
  for (int k1 = 0; k1 < 4; ++k1) {
    if (k1 == 2) fft4_r2<DIR>(a[0][k1], a[1][k1], a[2][k1], a[3][k1]);
    else fft4<DIR>(a[0][k1], a[1][k1], a[2][k1], a[3][k1]);   // -> k2 = 0..3
#pragma unroll
    for (int k2 = 0; k2 < 4; ++k2) v[k1 + 4 * k2] = a[k2][k1];
  }
}

}  // namespace icsfft

// ---- arguments ---------------------------------------------------------------------------------------------------------------------------
struct IcsFftArgs {
  IcsConvArgs c;        // frames, operands, reduction slots, geometry (c.w / c.bt / c.facc / c.sched unused)
  const v2f* spec;      // [3][128][128]: conj(DFT2(W_c)) / 128^2 of this orientation (k_fft_spectrum)
  const v2f* spec1;     // mode 2 (k_conv_fft<2>: A1 + A3 in one unit): `spec` = the convolution orientation's, `spec1` = the correlation orientation's
  int V;                // valid output pixels per tile ROW: 128 - K + 1 rounded down to whole quads
  int Vy;               // valid output ROWS per tile = 128 - K + 1: rows need no rounding to quads, and two more rows per tile save a whole round of
                        // units at some sizes (6144^2 / 31 x 31 back-projection: 65 x 65 tiles -> 63 x 65 = exactly 24 units per CU instead of 24.8)
  int tiles_x, ntiles, nunits;
  unsigned long long tiles_x_magic;   // floor(2^32 / tiles_x) + 1 (33 bits for tiles_x = 1): the unit decode divides by a multiply
  int oy0, ox0, oy1, ox1;   // output region in u-frame coordinates (mode 0: the M x N interior; mode 1: the whole u-frame)
  int gx0;                  // first column of the tile grid: ox0 rounded down to a multiple of 4, so that every 16-byte access of a plane row
                            // is 16-byte aligned (measured on MI355X: a buffer_store_dwordx4 at 12 mod 16 bytes lost its first dword on
                            // some lanes); the pixels in front of ox0 are stored as zeros, like those behind ox1
  int planar;               // bit mask of the frames that are channel-planar mirrors (ics_common.h): ICS_FFT_PL_*
  int wpad;                 // a tile's window starts wpad pixels up and left of its first output pixel: pad (one convolution), 2 pad (mode 2, k_conv_fft<2>: two in a row)
  float* fspec;             // mode 2: DFT of the image windows of every unit, [unit][8][1024] quads in load_spectrum's order (k_fft_image_spectrum)
  int blk_n, blk_k;         // tap blocks (PSF sizes above ICS_FFT_MAX_K, k_conv_fft_blk / k_gradk_fft with a lag block): blk_n x blk_n blocks of blk_k x blk_k taps;
                            // the tiles' valid part follows the BLOCK size, 128 - blk_k + 1 pixels a side.  0 = the whole PSF in one tile
  int lag_y, lag_x;         // k_gradk_fft with tap blocks: the block of lags [lag_y, lag_y + blk_k) x [lag_x, lag_x + blk_k) this launch evaluates
  int rot;                  // the walk starts `rot` units into the unit list and wraps around (mode 2: so that the last, partial round of units is
                            // not the bottom tile row, whose units are the outer ring's four-transform ones).  Order only: results do not change
  int wy0, wy1, wx0, wx1;   // k_synth_gradk_fft: the stop-test window in u-frame coordinates -- the residual is stored to its frame for the tiles
  int store_all;            // that touch it (pyx:600-601, 627 read nothing else of it), or for every tile (single stage)
  long long* trace;         // harness builds with -DICS_FFT_TRACE: [workgroup][unit round][wave][10] shader-clock stamps, else unused
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
    const int ty = (int)(((unsigned long long)(unsigned)ti * a.tiles_x_magic) >> 32), tx = ti - ty * a.tiles_x;   // ti / tiles_x (fill_args: exact for ti * tiles_x < 2^32; the magic is 2^32 + 1 for one tile column)
    u.oy[t] = a.oy0 + ty * a.Vy; u.ox[t] = a.gx0 + tx * a.V;
  }
  return u;
}

// walk position k -> unit (positions beyond the list stay beyond it: their accesses are dropped)
ICS_FFT_HD int walk_unit(const IcsFftArgs& a, int k) {
  if (k >= a.nunits) return k;
  const int n = k + a.rot;
  return n < a.nunits ? n : n - a.nunits;
}
ICS_FFT_HD Lay make_lay(const IcsGeom& g, bool) {
  Lay l;
  l.pitch = ics_ppitch(g); l.org = g.ay * l.pitch + g.ax; l.cmul = g.rows * l.pitch;
  return l;
}
// (mode = 0 / 1: only the frames that mode touches get a resource of their own -- scalar registers are short in this kernel; -1: all)
ICS_FFT_HD Mem make_mem(const IcsFftArgs& a, int mode = -1) {
  Mem m;
  const IcsGeom& g = a.c.g;
  m.lin = make_lay(g, a.planar & ICS_FFT_PL_IN); m.lout = make_lay(g, a.planar & ICS_FFT_PL_OUT); m.lf = make_lay(g, a.planar & ICS_FFT_PL_F);
  m.lu = make_lay(g, a.planar & ICS_FFT_PL_U); m.lut = make_lay(g, a.planar & ICS_FFT_PL_UT); m.ltv = make_lay(g, a.planar & ICS_FFT_PL_TV);
  m.in = make_gbuf(a.c.in - m.lin.org); m.out = make_gbuf(a.c.out - m.lout.org);
  m.f = mode == 1 ? m.in : make_gbuf(a.c.f - m.lf.org);
  m.u = (mode == 0 || mode == 2) ? m.in : make_gbuf(a.c.u - m.lu.org);    // (mode 2 convolves u itself: the window's frame is the operand frame)
  m.ut = mode == 0 ? m.in : make_gbuf(a.c.ut - m.lut.org);
  m.tv = (a.c.tv && mode != 0) ? make_gbuf(a.c.tv - m.ltv.org) : m.in;
  m.spec = make_gbuf(a.spec);
  m.spec1 = (mode == 2 || mode == -1) ? make_gbuf(a.spec1) : m.spec;
  m.fspec = (mode == 2 || mode == -1) ? make_gbuf(a.fspec) : m.spec;
  return m;
}

// LDS reads are volatile: left alone, the compiler pairs them into ds_read2_b64 / ds_read2st64_b64, which take 8 LDS cycles per wave
// instruction where two ds_read_b64 take 2 + 2 (MI355X_MICROARCH: 128 vs 256 B/clk)
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ v2f lds_ld(const v2f* p) {   // (the low half of a generic address inside the LDS aperture is the LDS address)
  typedef const volatile __attribute__((address_space(3))) v2f* lds_vp;
  return *(lds_vp)(uint32_t)(uintptr_t)p;
}
#else
static inline v2f lds_ld(const v2f* p) { return *p; }
#endif

// "everything requested above is issued before anything below": keeps the scheduler from sinking LDS reads next to their first use, which
// turns fifteen twiddle reads into fifteen serial LDS round trips (seen in the ISA of stage E: ds_read / s_waitcnt lgkmcnt(0) / multiply, x 15)
#if defined(__HIP_DEVICE_COMPILE__)
#define ICS_FFT_ISSUE_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define ICS_FFT_ISSUE_FENCE() do { } while (0)
#endif

// Thread mappings.  x-major (stages A, B, F, G): wave w -> selector w & 7, columns 64 (w >> 3) + lane.  Row-owner (C, D, E): wave w -> rows
// 8 w + (lane >> 3), selector lane & 7.  Row-quad (the unit's boundaries, 16-byte global accesses): rows (tid >> 5) + 32 i, i < 4, pixels
// 4 (tid & 31) .. + 3 -- a half-wave covers one 512-byte row segment.
#define ICS_FFT_AT(row, col) lds[(row) * ICS_FFT_PITCH + (col)]

// The window of a unit, requested one unit ahead into registers: tile t, row group i -> 4 consecutive pixels (one dwordx4; a wave64 memory
// instruction costs the texture addresser ~16 cycles whether it moves 4 or 16 bytes per lane: as single floats the 64 loads per thread of
// a unit took 20 k of its 37 k shader clocks).  Rows and pixels beyond the frame's value range read as 0: rows as dropped accesses, pixels
// through the apron's zeros (a quad that starts inside [.., uN + pad) ends inside the apron, ax >= pad + 3; quads beyond it are dropped).
// (dy, dx: the window starts that much further down / right -- the tap blocks of wide PSFs)
ICS_FFT_HD void load_window(const IcsFftArgs& a, const Mem& mem, const Unit& u, int tid, v4f (&pw)[2][4], int t0 = 0, int t1 = 2, int dy = 0, int dx = 0) {
  const int r0 = tid >> 5, xq = tid & 31;
  const int pad = a.wpad, pitch = mem.lin.pitch, ylast = a.c.g.uM + pad - 1, xlast = a.c.g.uN + pad - 1;
#pragma unroll
  for (int t = t0; t < t1; ++t) {
    const int X = u.ox[t] - pad + dx + 4 * xq, Y0 = u.oy[t] - pad + dy + r0;       // both >= -pad by construction
    const int vo = (u.has[t] && X <= xlast) ? mem.lin.org + Y0 * pitch + X + mem.lin.cmul * u.c : ICS_FFT_NONE;
#pragma unroll
    for (int i = 0; i < 4; ++i) pw[t][i] = ld_f32x4<4>(mem.in, (Y0 + 32 * i <= ylast) ? vo : ICS_FFT_NONE, 32 * i * pitch);
  }
}
// ... and its way into the tile buffer: z = tile 0 + i tile 1, natural [row][pixel] layout (two 16-byte LDS stores per row group)
ICS_FFT_HD void store_window(const v4f (&pw)[2][4], v2f* lds, int tid) {
  const int r0 = tid >> 5, xq = tid & 31;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    v4f* wp = reinterpret_cast<v4f*>(lds + (r0 + 32 * i) * ICS_FFT_PITCH + 4 * xq);
    wp[0] = (v4f){pw[0][i].x, pw[1][i].x, pw[0][i].y, pw[1][i].y};
    wp[1] = (v4f){pw[0][i].z, pw[1][i].z, pw[0][i].w, pw[1][i].w};
  }
}

// A: column x, rows j + 8 m -> radix-16 over m -> twiddle (wave-uniform: scalar registers) -> k1 to row j + 8 k1 (the slots it read)
ICS_FFT_HD void stage_a(v2f* lds, int tid) {
  const int w = ICS_FFT_UNIFORM(tid >> 6), lane = tid & 63, j = w & 7, x = 64 * (w >> 3) + lane;
  v2f* cp = lds + j * ICS_FFT_PITCH + x;
  v2f v[16];
#pragma unroll
  for (int m = 0; m < 16; ++m) v[m] = lds_ld(cp + 8 * m * ICS_FFT_PITCH);
  fft16<1>(v);
#pragma unroll
  for (int k1 = 0; k1 < 16; ++k1) cp[8 * k1 * ICS_FFT_PITCH] = k1 ? cmul_s(v[k1], tw128(j * k1)) : v[k1];
}

// B: radix-8 over j at fixed k1 (rows j + 8 k1) -> k2 to row 8 k1 + k2: frequency ky = k1 + 16 k2 lives in row 8 k1 + k2 from here on
// (in place again: the row stages do not care which row holds which ky, stage D asks ky_of_row).  F = the inverse, the same slots.
template <int DIR> ICS_FFT_HD void stage_b(v2f* lds, int tid) {
  const int w = ICS_FFT_UNIFORM(tid >> 6), lane = tid & 63, x = 64 * (w >> 3) + lane;
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    v2f* bp = lds + 8 * ((w & 7) + 8 * s) * ICS_FFT_PITCH + x;
    v2f v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = lds_ld(bp + i * ICS_FFT_PITCH);
    if (DIR > 0) fft8<1>(v); else fft8<-1>(v);
#pragma unroll
    for (int i = 0; i < 8; ++i) bp[i * ICS_FFT_PITCH] = v[i];
  }
}
ICS_FFT_HD int ky_of_row(int row) { return (row >> 3) + 16 * (row & 7); }

ICS_FFT_HD int skew_col(int j, int k1) { return 8 * k1 + ((j + k1) & 7); }

// C: one row, x = j + 8 m -> radix-16 over m -> twiddle -> column 8 k1 + (j + k1) % 8
// (`rd` = `lds` on the device -- the lanes of a wave run in lock step, every read is back before the first write; the CPU emulation, which
//  runs the threads one after the other, passes a snapshot)
// (`twl` = the 128 twiddles in LDS behind the tile: the lane-dependent ones of C and E are read from there, all fifteen requested ahead of
//  the transform; the skewed columns are eight base addresses (j + s) % 8, s = k1 % 8, plus compile-time offsets)
template <int TWB = 8>   // twiddles requested TWB at a time (8: two halves; 4: the PSF-gradient kernel, which holds 64 registers of spectra beside this stage)
ICS_FFT_HD void stage_c(const v2f* rd, v2f* lds, const v2f* twl, int tid) {
  const int w = ICS_FFT_UNIFORM(tid >> 6), lane = tid & 63, row = 8 * w + (lane >> 3), j = lane & 7;
  v2f v[16], tw[TWB];
  const v2f* rp = rd + row * ICS_FFT_PITCH + j;
#pragma unroll
  for (int m = 0; m < 16; ++m) v[m] = lds_ld(rp + 8 * m);
  if (TWB == 8) {
#pragma unroll
    for (int k1 = 1; k1 < 8; ++k1) tw[k1] = lds_ld(twl + j * ICS_FFT_TWS + k1);
    ICS_FFT_ISSUE_FENCE();
  }
  fft16<1>(v);
  v2f* const rowp = lds + row * ICS_FFT_PITCH;
  if (TWB == 8) {      // the second eight twiddles are requested before the first eight products: those cover their round trip
    v2f tw2[8];
#pragma unroll
    for (int k1 = 8; k1 < 16; ++k1) tw2[k1 - 8] = lds_ld(twl + j * ICS_FFT_TWS + k1);
    ICS_FFT_ISSUE_FENCE();
#pragma unroll
    for (int k1 = 0; k1 < 8; ++k1) rowp[8 * k1 + ((j + k1) & 7)] = k1 ? cmul(v[k1], tw[k1]) : v[k1];
#pragma unroll
    for (int k1 = 8; k1 < 16; ++k1) rowp[8 * k1 + ((j + k1) & 7)] = cmul(v[k1], tw2[k1 - 8]);
    return;
  }
#pragma unroll
  for (int h = 0; h < 16 / TWB; ++h) {
    if (TWB != 8 || h > 0) {
#pragma unroll
      for (int k1 = TWB * h; k1 < TWB * h + TWB; ++k1) if (k1) tw[k1 - TWB * h] = lds_ld(twl + j * ICS_FFT_TWS + k1);
    }
#pragma unroll
    for (int k1 = TWB * h; k1 < TWB * h + TWB; ++k1) rowp[8 * k1 + ((j + k1) & 7)] = k1 ? cmul(v[k1], tw[k1 - TWB * h]) : v[k1];
  }
}

// The spectrum values a thread multiplies by in stage D: row -> ky, kx = q + 8 s + 16 k2 (from L2: 384 KB for the three channels).  Requested
// in front of stage C: requested inside stage D, in two batches of eight with a wait each, the last wave left stage D 17 k shader clocks
// after the first (per-wave timeline, tools/bench_conv_fft.hip -DICS_FFT_TRACE).  (Requested a whole unit ahead -- in front of the
// previous unit's stores, which vmcnt makes every later load wait for -- they took stage D to 2 k clocks, but 32 registers alive across
// stages A-C spilled the window prefetch: measured slower.)
// Layout (k_fft_spectrum writes it): the sixteen values of a thread as eight 16-byte pairs, [channel][pair l = 4 s + k2 / 2][thread] -- a wave's
// load is 1 KiB contiguous, eight loads per thread instead of sixteen 8-byte ones (the texture addresser's time goes by instructions).
ICS_FFT_HD int spec_index(int c, int ky, int kx) {   // position of S_c[ky][kx] in v2f units
  const int tid = 64 * (ky & 15) + 8 * (ky >> 4) + (kx & 7), s = (kx >> 3) & 1, k2 = kx >> 4;
  return (((c * 8 + 4 * s + (k2 >> 1)) * ICS_FFT_THREADS + tid) * 2) + (k2 & 1);
}
ICS_FFT_HD void load_spectrum(const Mem& mem, int c, int tid, v2f (&sp)[2][8]) {
#pragma unroll
  for (int l = 0; l < 8; ++l) {
    const v4f p = ld_f32x4<1>(mem.spec, 4 * tid, (c * 8 + l) * ICS_FFT_THREADS * 4);
    sp[l >> 2][2 * (l & 3)] = (v2f){p.x, p.y};
    sp[l >> 2][2 * (l & 3) + 1] = (v2f){p.z, p.w};
  }
}
// one half (s = 0 / 1: the eight values of one pass of stage D) of a thread's sixteen spectrum values, from a buffer in load_spectrum's layout
// whose block of 8 x 1024 quads starts at quad index `blk` (a channel of the weight spectra, a unit of the image spectra)
template <int KIND>
ICS_FFT_HD void load_spectrum_half(gbuf b, int blk, int tid, int s, v2f (&sp)[8]) {
#pragma unroll
  for (int l = 0; l < 4; ++l) {
    const v4f p = ld_f32x4<KIND>(b, 4 * tid, (blk + 4 * s + l) * ICS_FFT_THREADS * 4);
    sp[2 * l] = (v2f){p.x, p.y};
    sp[2 * l + 1] = (v2f){p.z, p.w};
  }
}
// ... and the way out: a thread's sixteen values as block `blk` of such a buffer (k_fft_image_spectrum)
ICS_FFT_HD void store_spectrum(gbuf b, int blk, int tid, const v2f (&z)[2][8]) {
#pragma unroll
  for (int l = 0; l < 8; ++l) {
    const v2f z0 = z[l >> 2][2 * (l & 3)], z1 = z[l >> 2][2 * (l & 3) + 1];
    st_f32x4(b, 4 * tid, (blk + l) * ICS_FFT_THREADS * 4, (v4f){z0.x, z0.y, z1.x, z1.y});
  }
}
// Stage D of the fused A1 + A3 unit (mode 2, k_conv_fft<2>), interior tiles: with T = the window's spectrum (after the radix-8 pass),
//     G = S1 . (16384 S0 T - F),     F = the UNNORMALISED transform of the image window (k_fft_image_spectrum),
// i.e. the spectrum of corr(conv(u) - image): both weight spectra carry the 1 / 128^2 of an inverse transform, the first one's is undone
// (a power of two: exact).  One forward and one inverse transform where k_conv_fft<0> + k_conv_fft<1> run two of each.
ICS_FFT_HD void stage_d2_half(const v2f (&s0)[8], const v2f (&s1)[8], const v2f (&fs)[8], v2f* lds, int tid, int s) {
  const int w = ICS_FFT_UNIFORM(tid >> 6), lane = tid & 63, row = 8 * w + (lane >> 3), q = lane & 7;
  v2f v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = lds_ld(lds + row * ICS_FFT_PITCH + 8 * q + ((j + q) & 7) + 64 * s);
  fft8<1>(v);
#pragma unroll
  for (int k2 = 0; k2 < 8; ++k2) {
    const v2f st = cmul(v[k2], s0[k2]);
    const v2f x = __builtin_elementwise_fma(st, (v2f){16384.f, 16384.f}, -fs[k2]);
    v[k2] = cmul(x, s1[k2]);
  }
  fft8<-1>(v);
#pragma unroll
  for (int j = 0; j < 8; ++j) lds[row * ICS_FFT_PITCH + 8 * q + ((j + q) & 7) + 64 * s] = v[j];
}

// D: radix-8 over j -> kx = k1 + 16 k2, multiply by the spectrum, inverse radix-8 over k2 -> j, same slots
ICS_FFT_HD void stage_d(const v2f (&sp)[2][8], v2f* lds, int tid) {
  const int w = ICS_FFT_UNIFORM(tid >> 6), lane = tid & 63, row = 8 * w + (lane >> 3), q = lane & 7;
  v2f* db[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) db[j] = lds + row * ICS_FFT_PITCH + 8 * q + ((j + q) & 7);   // column 8 k1 + (j + k1) % 8 with k1 = q + 8 s: + 64 s
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    v2f v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = lds_ld(db[j] + 64 * s);
    fft8<1>(v);
#pragma unroll
    for (int k2 = 0; k2 < 8; ++k2) v[k2] = cmul(v[k2], sp[s][k2]);
    fft8<-1>(v);
#pragma unroll
    for (int j = 0; j < 8; ++j) db[j][64 * s] = v[j];
  }
}

// The two halves of stage D on their own (PSF gradient, k_gradk_fft): the 2-D spectrum of the tile in the thread's registers -- sixteen
// values, the same (ky, kx) in the same slot for every tile -- and the way back from such a set of values.
ICS_FFT_HD void stage_d_forward(const v2f* lds, int tid, v2f (&z)[2][8]) {
  const int w = ICS_FFT_UNIFORM(tid >> 6), lane = tid & 63, row = 8 * w + (lane >> 3), q = lane & 7;
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    v2f v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = lds_ld(lds + row * ICS_FFT_PITCH + 8 * q + ((j + q) & 7) + 64 * s);
    fft8<1>(v);
#pragma unroll
    for (int k2 = 0; k2 < 8; ++k2) z[s][k2] = v[k2];
  }
}
ICS_FFT_HD void stage_d_inverse(const v2f (&z)[2][8], v2f* lds, int tid) {
  const int w = ICS_FFT_UNIFORM(tid >> 6), lane = tid & 63, row = 8 * w + (lane >> 3), q = lane & 7;
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    v2f v[8];
#pragma unroll
    for (int k2 = 0; k2 < 8; ++k2) v[k2] = z[s][k2];
    fft8<-1>(v);
#pragma unroll
    for (int j = 0; j < 8; ++j) lds[row * ICS_FFT_PITCH + 8 * q + ((j + q) & 7) + 64 * s] = v[j];
  }
}

// Stage D of the fused A11 + A13 unit (k_synth_gradk_fft).  First use: as stage_d, and the window's 2-D spectrum stays behind in `zu`
// (the same (ky, kx) in the same slot for every tile: stage_d_forward's layout).  Second use, on the residual tile: its spectrum goes
// straight into the workgroup's sum  acc += DFT(t) conj(DFT(e'))  -- eight values at a time, the residual's spectrum is never whole in registers.
ICS_FFT_HD void stage_d_keep(const v2f (&sp)[2][8], v2f* lds, int tid, v2f (&zu)[2][8]) {
  const int w = ICS_FFT_UNIFORM(tid >> 6), lane = tid & 63, row = 8 * w + (lane >> 3), q = lane & 7;
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    v2f v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = lds_ld(lds + row * ICS_FFT_PITCH + 8 * q + ((j + q) & 7) + 64 * s);
    fft8<1>(v);
#pragma unroll
    for (int k2 = 0; k2 < 8; ++k2) { zu[s][k2] = v[k2]; v[k2] = cmul(v[k2], sp[s][k2]); }
    fft8<-1>(v);
#pragma unroll
    for (int j = 0; j < 8; ++j) lds[row * ICS_FFT_PITCH + 8 * q + ((j + q) & 7) + 64 * s] = v[j];
  }
}
ICS_FFT_HD void stage_d_acc(const v2f* lds, int tid, const v2f (&zu)[2][8], v2f (&acc)[2][8]) {
  const int w = ICS_FFT_UNIFORM(tid >> 6), lane = tid & 63, row = 8 * w + (lane >> 3), q = lane & 7;
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    v2f v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = lds_ld(lds + row * ICS_FFT_PITCH + 8 * q + ((j + q) & 7) + 64 * s);
    fft8<1>(v);
#pragma unroll
    for (int k2 = 0; k2 < 8; ++k2) acc[s][k2] += cmulc(zu[s][k2], v[k2]);
  }
}

// Tap blocks (k_conv_fft_blk): the window's spectrum times the block's weight spectrum, added to the unit's sum -- the products of all
// blocks meet in the frequency domain and share one inverse transform (stage_d_inverse)
ICS_FFT_HD void stage_d_mac_half(const v2f* lds, int tid, int s, const v2f (&sp)[8], v2f (&acc)[8]) {
  const int w = ICS_FFT_UNIFORM(tid >> 6), lane = tid & 63, row = 8 * w + (lane >> 3), q = lane & 7;
  v2f v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = lds_ld(lds + row * ICS_FFT_PITCH + 8 * q + ((j + q) & 7) + 64 * s);
  fft8<1>(v);
#pragma unroll
  for (int k2 = 0; k2 < 8; ++k2) acc[k2] += cmul(v[k2], sp[k2]);
}
ICS_FFT_HD void stage_d_mac(const v2f* lds, int tid, const v2f (&sp)[2][8], v2f (&acc)[2][8]) {
  stage_d_mac_half(lds, tid, 0, sp[0], acc[0]);
  stage_d_mac_half(lds, tid, 1, sp[1], acc[1]);
}

// E with its twiddles requested four at a time (the fused unit holds 64 registers of spectra beside this stage: as stage_c<4>)
ICS_FFT_HD void stage_e_lean(const v2f* rd, v2f* lds, const v2f* twl, int tid) {
  const int w = ICS_FFT_UNIFORM(tid >> 6), lane = tid & 63, row = 8 * w + (lane >> 3), j = lane & 7;
  v2f v[16];
#pragma unroll
  for (int k1 = 0; k1 < 16; ++k1) v[k1] = lds_ld(rd + row * ICS_FFT_PITCH + ((j + k1) & 7) + 8 * k1);
#pragma unroll
  for (int h = 0; h < 4; ++h) {
    v2f tw[4];
#pragma unroll
    for (int k1 = 4 * h; k1 < 4 * h + 4; ++k1) if (k1) tw[k1 - 4 * h] = lds_ld(twl + j * ICS_FFT_TWS + k1);
    ICS_FFT_ISSUE_FENCE();
#pragma unroll
    for (int k1 = 4 * h; k1 < 4 * h + 4; ++k1) if (k1) v[k1] = cmulc(v[k1], tw[k1 - 4 * h]);
  }
  fft16<-1>(v);
  v2f* wp = lds + row * ICS_FFT_PITCH + j;
#pragma unroll
  for (int m = 0; m < 16; ++m) wp[8 * m] = v[m];
}

// E: conj twiddle, inverse radix-16 over k1 -> x = j + 8 m
ICS_FFT_HD void stage_e(const v2f* rd, v2f* lds, const v2f* twl, int tid) {
  const int w = ICS_FFT_UNIFORM(tid >> 6), lane = tid & 63, row = 8 * w + (lane >> 3), j = lane & 7;
  v2f v[16], tw[16];
  const v2f* cb[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) cb[s] = rd + row * ICS_FFT_PITCH + ((j + s) & 7);
  // all fifteen twiddles and the sixteen values requested in one go (the scheduler otherwise sinks each twiddle read next to its product:
  // fifteen serial LDS round trips per wave in a stage every wave of the CU is in at the same time)
#pragma unroll
  for (int k1 = 1; k1 < 16; ++k1) tw[k1] = lds_ld(twl + j * ICS_FFT_TWS + k1);
#pragma unroll
  for (int k1 = 0; k1 < 16; ++k1) v[k1] = lds_ld(cb[k1 & 7] + 8 * k1);
  ICS_FFT_ISSUE_FENCE();
#pragma unroll
  for (int k1 = 1; k1 < 16; ++k1) v[k1] = cmulc(v[k1], tw[k1]);
  fft16<-1>(v);
  v2f* wp = lds + row * ICS_FFT_PITCH + j;
#pragma unroll
  for (int m = 0; m < 16; ++m) wp[8 * m] = v[m];
}

// G: rows j + 8 k1 of column x -> conj twiddle, inverse radix-16 over k1 -> the finished values of rows j + 8 m, back into the slots they
// came from: the tile buffer now holds r (tile 0 in .x, tile 1 in .y) in natural [row][pixel] layout for the row-quad epilogue
ICS_FFT_HD void stage_g(v2f* lds, int tid) {
  const int w = ICS_FFT_UNIFORM(tid >> 6), lane = tid & 63, j = w & 7, x = 64 * (w >> 3) + lane;
  v2f* cp = lds + j * ICS_FFT_PITCH + x;
  v2f v[16];
#pragma unroll
  for (int k1 = 0; k1 < 16; ++k1) v[k1] = lds_ld(cp + 8 * k1 * ICS_FFT_PITCH);
#pragma unroll
  for (int k1 = 1; k1 < 16; ++k1) v[k1] = cmulc_s(v[k1], tw128(j * k1));
  fft16<-1>(v);
#pragma unroll
  for (int m = 0; m < 16; ++m) cp[8 * m * ICS_FFT_PITCH] = v[m];
}

// canonical positive NaN so that a NaN propagates through the integer max like np.amax does (ics_conv.hip)
ICS_FFT_HD uint32_t key_of(float f) { return (f != f) ? 0xFFC00000u : ics_f2key(f); }

// Epilogue (row-quad ownership): the arithmetic of ics_conv.hip on 4 consecutive pixels of a row at a time, operands and results as
// dwordx4.  On gfx9 vmcnt counts loads and stores alike and retires them in order: a load issued behind a store waits out the store's round
// trip to L2.  So within a unit every operand load is issued before the first store: mode 0 requests the image quads of both tiles before
// stage F; mode 1 walks its operands (u, ut[, T]) row group by row group, group i + 1 requested before the maxima of group i are taken, and
// stores the values it kept at the end.  A quad's lane address says "this row group of this tile is mine" or is a dropped access; pixels
// of a valid quad beyond the output region are stored as zeros (they land in the frame's border ring / slack, which holds zeros).
struct Ops { v4f a[2][4], b[2][4]; };   // [tile][row group].  mode 1: a = u, b = ut -- or, for the PAM kinds (TV kernel, tv_kind >= 2), b = the T frame
// The maxima of A6 / A7 over a unit's valid pixels, in a form that costs two or three vector operations per pixel and no lane masks:
//   ag  = max over pixels of (bits of g) & 0x7FFFFFFF as an unsigned integer: the bits of |g| order like |g| itself and every NaN lies above
//         +inf (0x7F800000), so one integer maximum carries both max |g| and "a NaN was seen";
//   mu  = float maximum of u (v_max_f32 drops NaNs), au = the same integer maximum of |u| bits, kept only for its NaN test;
//   any = a valid pixel was seen (row groups outside the tile / region contribute nothing).
struct Maxima { uint32_t ag, au, any; float mu; };
ICS_FFT_HD void maxima_init(Maxima& mx) { mx.ag = 0u; mx.au = 0u; mx.any = 0u; mx.mu = -__builtin_inff(); }
ICS_FFT_HD uint32_t fbits(float f) { return __builtin_bit_cast(uint32_t, f); }
// the unit's two keys (0 = nothing seen, canonical NaN key = largest: a NaN propagates like np.amax)
ICS_FFT_HD void maxima_keys(const Maxima& mx, uint32_t& kg, uint32_t& ku) {
  kg = mx.ag > 0x7F800000u ? 0xFFC00000u : (mx.any ? ics_f2key(__builtin_bit_cast(float, mx.ag)) : 0u);
  ku = mx.au > 0x7F800000u ? 0xFFC00000u : (mx.any ? ics_f2key(mx.mu) : 0u);
}

// lane address of row group 0 of tile t in frame layout L, or ICS_FFT_NONE; `rows` = number of this lane's row groups inside the tile (0..4)
ICS_FFT_HD int quad_lane(const IcsFftArgs& a, const Unit& u, const Lay& L, int tid, int t, int& rows, int& X) {
  const int r0 = tid >> 5, xq = tid & 31;
  const int lim = a.oy1 - u.oy[t] < a.Vy ? a.oy1 - u.oy[t] : a.Vy;      // output rows of this tile
  X = u.ox[t] + 4 * xq;
  const bool ok = u.has[t] && 4 * xq < a.V && X < a.ox1 && r0 < lim;
  rows = ok ? (lim - r0 + 31) >> 5 : 0;                                // row groups i with r0 + 32 i < lim
  return ok ? L.org + (u.oy[t] + r0) * L.pitch + X + L.cmul * u.c : ICS_FFT_NONE;
}

ICS_FFT_HD void load_image(const IcsFftArgs& a, const Mem& mem, const Unit& u, int tid, v4f (&f)[2][4], int t0 = 0, int t1 = 2) {
#pragma unroll
  for (int t = t0; t < t1; ++t) {
    int rows, X;
    const int vo = quad_lane(a, u, mem.lf, tid, t, rows, X);
#pragma unroll
    for (int i = 0; i < 4; ++i) f[t][i] = ld_f32x4<2>(mem.f, i < rows ? vo : ICS_FFT_NONE, 32 * i * mem.lf.pitch);
  }
}
// (row groups [i0, i1) of both tiles only)
ICS_FFT_HD void load_image_rows(const IcsFftArgs& a, const Mem& mem, const Unit& u, int tid, v4f (&f)[2][4], int i0, int i1) {
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    int rows, X;
    const int vo = quad_lane(a, u, mem.lf, tid, t, rows, X);
#pragma unroll
    for (int i = i0; i < i1; ++i) f[t][i] = ld_f32x4<2>(mem.f, i < rows ? vo : ICS_FFT_NONE, 32 * i * mem.lf.pitch);
  }
}
// mode 1: the operands under tile t.  The PAM kinds (build-defined tv_mode 2 / 3; ics_conv.hip's epilogue for them) need u and the TV
// term T = -div(p) instead of u and ut: two operand frames either way (a third does not fit 128 registers).
template <bool TV>
ICS_FFT_HD void load_ops(const IcsFftArgs& a, const Mem& mem, const Unit& u, int tid, int t, Ops& o, int i0 = 0, int i1 = 4) {
  int rows, X;
  const int va = quad_lane(a, u, mem.lu, tid, t, rows, X);   // (the same geometry: all frames of a job are)
  const bool pam = TV && a.c.tv_kind >= 2;
#pragma unroll
  for (int i = i0; i < i1; ++i) {
    const int vo = i < rows ? va : ICS_FFT_NONE;
    if (t == 0) {   // (ablation kinds: 2 = tile 0's operands and mode 0's image, 16 = tile 1's)
      o.a[t][i] = ld_f32x4<2>(mem.u, vo, 32 * i * mem.lu.pitch);
      o.b[t][i] = pam ? ld_f32x4<2>(mem.tv, vo, 32 * i * mem.ltv.pitch) : ld_f32x4<2>(mem.ut, vo, 32 * i * mem.lut.pitch);
    } else {
      o.a[t][i] = ld_f32x4<16>(mem.u, vo, 32 * i * mem.lu.pitch);
      o.b[t][i] = pam ? ld_f32x4<16>(mem.tv, vo, 32 * i * mem.ltv.pitch) : ld_f32x4<16>(mem.ut, vo, 32 * i * mem.lut.pitch);
    }
  }
}
// the finished values of row group i: r[t] = 4 pixels of tile t
ICS_FFT_HD void read_quads(const v2f* lds, int tid, int i, v4f (&r)[2]) {
  const int r0 = tid >> 5, xq = tid & 31;
  const v4f* rp = reinterpret_cast<const v4f*>(lds + (r0 + 32 * i) * ICS_FFT_PITCH + 4 * xq);
  const v4f z0 = rp[0], z1 = rp[1];
  r[0] = (v4f){z0.x, z0.z, z1.x, z1.z};
  r[1] = (v4f){z0.y, z0.w, z1.y, z1.w};
}
// mode 0 takes the lane address and the pixel masks of a tile ONCE per unit (as eight store_quad calls the address arithmetic of the
// epilogue was 300 of a unit's 1310 vector instructions); `edge` (wave-uniform) = the tile reaches beyond the output region's columns
struct QuadOut { int vo, rows, X; };
ICS_FFT_HD void store_quad_at(const IcsFftArgs& a, const Mem& mem, const QuadOut& q, bool edge, int i, v4f val) {
  if (edge) {
    const int X = q.X;
    val = (v4f){X >= a.ox0 ? val.x : 0.f, (X + 1 >= a.ox0 && X + 1 < a.ox1) ? val.y : 0.f, (X + 2 >= a.ox0 && X + 2 < a.ox1) ? val.z : 0.f, X + 3 < a.ox1 ? val.w : 0.f};
  }
  st_f32x4(mem.out, i < q.rows ? q.vo : ICS_FFT_NONE, 32 * i * mem.lout.pitch, val);
}
// Fused A11 + A13 unit: the residual of row group i, e' = r - image (pyx:563-565) on the tile's valid pixels inside the M x N interior and
// exact zeros everywhere else of the 128 x 128 tile (what k_gradk_fft reads back from the residual frame), goes back into the slots it
// was read from -- the operand of the second forward transform -- and, for tiles under the stop-test window, to the residual frame.
ICS_FFT_HD void residual_quads(const IcsFftArgs& a, const Mem& mem, const QuadOut (&qo)[2], bool edge, bool store, v2f* lds, int tid, int i, const v4f (&fimg)[2][4]) {
  v4f r[2];
  read_quads(lds, tid, i, r);
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const bool row_ok = i < qo[t].rows;
    const int X = qo[t].X;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float d = ICS_FSUB(r[t][e], fimg[t][i][e]);        // pyx:565
      const bool ok = edge ? (row_ok && X + e >= a.ox0 && X + e < a.ox1) : row_ok;
      r[t][e] = ok ? d : 0.f;
    }
  }
  const int r0 = tid >> 5, xq = tid & 31;
  v4f* wp = reinterpret_cast<v4f*>(lds + (r0 + 32 * i) * ICS_FFT_PITCH + 4 * xq);
  wp[0] = (v4f){r[0].x, r[1].x, r[0].y, r[1].y};
  wp[1] = (v4f){r[0].z, r[1].z, r[0].w, r[1].w};
  if (store) {
#pragma unroll
    for (int t = 0; t < 2; ++t) st_f32x4(mem.out, i < qo[t].rows ? qo[t].vo : ICS_FFT_NONE, 32 * i * mem.lout.pitch, r[t]);
  }
}

// ---- mode 2 (k_conv_fft<2>): A1 + A2 + A3 of a tile pair in one unit -----------------------------------------------------------------------
// The residual a back-projection tile reads is the M x N interior's (zero outside, pyx:482-491).  A tile whose residual window -- the
// V2 + 2 pad pixels a side around it -- lies inside the interior needs no mask and runs in the frequency domain alone (stage_d2_half); the
// tiles of the outer ring take both transforms pairs, with the mask in between (`border`).  u-frame coordinates.
ICS_FFT_HD bool tile_is_border(const IcsFftArgs& a, int oy, int ox) {
  const IcsGeom& g = a.c.g;
  return oy - g.pad < g.pad || oy + a.Vy + g.pad > g.pad + g.M || ox - g.pad < g.pad || ox + a.V + g.pad > g.pad + g.N;
}
ICS_FFT_HD bool unit_is_border(const IcsFftArgs& a, const Unit& u) {
  return tile_is_border(a, u.oy[0], u.ox[0]) || (u.has[1] && tile_is_border(a, u.oy[1], u.ox[1]));
}
// border units, between the two transform pairs: the tile buffer holds conv(u) of the window that starts (pad, pad) before the output
// tile; e = conv - image inside the interior, 0 outside it (pyx:488 and the zero extension of mode "full", pyx:491), back into the slots
// it was read from.  Row-quad ownership; the image quads of all four row groups are requested first.
ICS_FFT_HD void residual_window(const IcsFftArgs& a, const Mem& mem, const Unit& u, v2f* lds, int tid) {
  const IcsGeom& g = a.c.g;
  const int r0 = tid >> 5, xq = tid & 31, pad = g.pad;
  v4f f[2][4];
  int X[2], Y0[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    X[t] = u.ox[t] - pad + 4 * xq; Y0[t] = u.oy[t] - pad + r0;
    const int vo = (u.has[t] && X[t] + 3 >= pad && X[t] < pad + g.N) ? mem.lf.org + Y0[t] * mem.lf.pitch + X[t] + mem.lf.cmul * u.c : ICS_FFT_NONE;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int Y = Y0[t] + 32 * i;
      f[t][i] = ld_f32x4<2>(mem.f, (Y >= pad && Y < pad + g.M) ? vo : ICS_FFT_NONE, 32 * i * mem.lf.pitch);
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    v4f r[2];
    read_quads(lds, tid, i, r);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int Y = Y0[t] + 32 * i;
      const bool row_ok = u.has[t] && Y >= pad && Y < pad + g.M;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float d = ICS_FSUB(r[t][e], f[t][i][e]);                 // pyx:488
        r[t][e] = (row_ok && X[t] + e >= pad && X[t] + e < pad + g.N) ? d : 0.f;
      }
    }
    v4f* wp = reinterpret_cast<v4f*>(lds + (r0 + 32 * i) * ICS_FFT_PITCH + 4 * xq);
    wp[0] = (v4f){r[0].x, r[1].x, r[0].y, r[1].y};
    wp[1] = (v4f){r[0].z, r[1].z, r[0].w, r[1].w};
  }
}

// mode 1: g = lambd gradu + (u - ut)/2 (pyx:519) for the maxima of A7 on row group i of tile t; the PAM kinds replace the stored value by G
template <bool TV>
ICS_FFT_HD void maxima_quad(const IcsFftArgs& a, const Unit& u, int tid, int t, int i, v4f& r, const Ops& o, Maxima& mx, const QuadOut& q, bool edge) {
  const float lambd = a.c.lambd;
  const int X0 = q.X;
  const bool row_ok = i < q.rows;         // (quad_lane: tile present, quad inside the tile's valid columns and the region, row group inside)
  uint32_t qg = 0u, qu = 0u, qany = 0u;
  float qm = -__builtin_inff();
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float rv = r[e], uv = o.a[t][i][e], tv = o.b[t][i][e];
    const int X = X0 + e;
    float g;
    if (TV && a.c.tv_kind >= 2) { g = (float)((double)tv + (double)ICS_FMUL(lambd, rv)); r[e] = g; }     // PAM: G = T + lambd*gradu, stored (o.b holds T)
    else
      g = ICS_FADD(ICS_FMUL(lambd, rv), ICS_FMUL(ICS_FSUB(uv, tv), 0.5f));                                          // pyx:519
    if (edge) {                                        // (wave-uniform) first / last tile of a tile row: per-pixel column test
      const bool ok = row_ok && X >= a.ox0 && X < a.ox1;
      qg = __builtin_elementwise_max(qg, ok ? (fbits(g) & 0x7FFFFFFFu) : 0u);
      qu = __builtin_elementwise_max(qu, ok ? (fbits(uv) & 0x7FFFFFFFu) : 0u);
      qm = __builtin_fmaxf(qm, ok ? uv : -__builtin_inff());
      qany |= ok ? 1u : 0u;
    } else {
      qg = __builtin_elementwise_max(qg, fbits(g) & 0x7FFFFFFFu);
      qu = __builtin_elementwise_max(qu, fbits(uv) & 0x7FFFFFFFu);
      qm = __builtin_fmaxf(qm, uv);
    }
  }
  if (edge) { mx.ag = __builtin_elementwise_max(mx.ag, qg); mx.au = __builtin_elementwise_max(mx.au, qu); mx.mu = __builtin_fmaxf(mx.mu, qm); mx.any |= qany; }
  else {
    mx.ag = __builtin_elementwise_max(mx.ag, row_ok ? qg : 0u); mx.au = __builtin_elementwise_max(mx.au, row_ok ? qu : 0u);
    mx.mu = __builtin_fmaxf(mx.mu, row_ok ? qm : -__builtin_inff()); mx.any |= row_ok ? 1u : 0u;
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

// a copy of `x` the optimiser cannot trace back: lane constants derived from it (LDS addresses, frame offsets) are recomputed in the stage that
// uses them instead of being hoisted out of the unit loop and kept alive -- and spilled -- across it (as in ics_conv_mfma.hip)
__device__ __forceinline__ int opaque(int x) { asm volatile("" : "+v"(x)); return x; }

// workgroup barrier that waits for this wave's LDS traffic only (__syncthreads() also waits for the global loads and stores in flight)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

#ifndef ICS_FFT_M2_ORDER
#define ICS_FFT_M2_ORDER 1   /* mode 2: 0 = the second halves of the weight spectra requested in front of stage D's first pass, 1 = behind it */
#endif
#ifndef ICS_FFT_M1_EARLY
#define ICS_FFT_M1_EARLY 1      // mode 1: row groups of tile 1's operands requested before stage F already (0: all at the start of the epilogue)
#endif
#ifndef ICS_FFT_M1_WIN
#define ICS_FFT_M1_WIN 0        // mode 1: tile 1 of the next window requested at the start of the epilogue instead of between its passes
#endif
template <int MODE, bool TV>
__global__ __launch_bounds__(ICS_FFT_THREADS) void k_conv_fft(IcsFftArgs a) {
  extern __shared__ __attribute__((aligned(16))) v2f lds[];
  v2f* const twl = lds + ICS_FFT_P * ICS_FFT_PITCH;
  const int tid = threadIdx.x;
  const int G = gridDim.x;
  if (tid < ICS_FFT_TW_ENTRIES) twl[tid] = tw128((tid / ICS_FFT_TWS) * (tid % ICS_FFT_TWS));
  const Mem mem = make_mem(a, MODE);
  // workgroup b runs on XCD b % 8 (observed dispatch): consecutive unit slots q go to one XCD, so the three channel units of a tile pair
  // (n = 3 pair + c) share that XCD's L2.  Affects speed only.
  const int q = (G & 7) ? (int)blockIdx.x : (int)(blockIdx.x & 7) * (G >> 3) + (int)(blockIdx.x >> 3);
  ICS_FFT_PROBE_TRACE_DECL();
  uint32_t accg[3] = {0u, 0u, 0u}, accu[3] = {0u, 0u, 0u};   // the workgroup's maxima as order-preserving keys (0 = nothing seen, NaN = largest)
  // A unit's window is requested one unit ahead (registers).  It enters the tile buffer -- and runs its stage A -- at the END of the unit
  // before it, behind that unit's stores: there the compiler knows exactly what is in flight (the window loads, then the stores) and waits
  // with vmcnt(n_stores).  (Consumed at the top of the loop the wait became vmcnt(0): the loop header merges the first entry, where
  // nothing follows the loads.)
  v4f pw[2][4];
  ICS_FFT_PROBE_STAGGER();
  if (q < a.nunits) {
    load_window(a, mem, decode_unit(a, walk_unit(a, q)), opaque(tid), pw);
    store_window(pw, lds, opaque(tid));
    lds_barrier();
    stage_a(lds, opaque(tid));
  }
  for (int k = q; k < a.nunits; k += G) {
    const int n = walk_unit(a, k);
    const Unit u = decode_unit(a, n);
    ICS_FFT_STAMP(0);
    lds_barrier();
    ICS_FFT_STAMP(1);
    ICS_FFT_PROBE_COLUMN_PASS(stage_b<1>(lds, opaque(tid)); lds_barrier(););
    ICS_FFT_STAMP(2);
    if (MODE == 2) {
      // A1 + A3 in one unit (see stage_d2_half): interior tiles stay in the frequency domain between the two convolutions
      if (!unit_is_border(a, u)) {
        v2f fs[2][8], s0[8], s1[8];
        load_spectrum_half<1>(mem.fspec, 8 * n, opaque(tid), 0, fs[0]);       // the image windows' spectrum of this unit: from HBM, requested first
        load_spectrum_half<1>(mem.fspec, 8 * n, opaque(tid), 1, fs[1]);
        load_spectrum_half<1>(mem.spec, 8 * u.c, opaque(tid), 0, s0);
        load_spectrum_half<1>(mem.spec1, 8 * u.c, opaque(tid), 0, s1);
        stage_c<4>(lds, lds, twl, opaque(tid));
        wave_sync();
#if ICS_FFT_M2_ORDER == 0
        v2f s0b[8], s1b[8];                                                  // the second halves of the weight spectra (L2) behind the first pass of stage D
        load_spectrum_half<1>(mem.spec, 8 * u.c, opaque(tid), 1, s0b);
        load_spectrum_half<1>(mem.spec1, 8 * u.c, opaque(tid), 1, s1b);
        stage_d2_half(s0, s1, fs[0], lds, opaque(tid), 0);
        stage_d2_half(s0b, s1b, fs[1], lds, opaque(tid), 1);
#else
        stage_d2_half(s0, s1, fs[0], lds, opaque(tid), 0);
        load_spectrum_half<1>(mem.spec, 8 * u.c, opaque(tid), 1, s0);
        load_spectrum_half<1>(mem.spec1, 8 * u.c, opaque(tid), 1, s1);
        stage_d2_half(s0, s1, fs[1], lds, opaque(tid), 1);
#endif
        wave_sync();
      } else {
        // the outer ring: conv, residual with its mask in the tile buffer, then the correlation (four transforms, as the two kernels)
        {
          v2f sp[2][8];
          load_spectrum(mem, u.c, opaque(tid), sp);
          stage_c(lds, lds, twl, opaque(tid));
          wave_sync();
          stage_d(sp, lds, opaque(tid));
        }
        wave_sync();
        stage_e(lds, lds, twl, opaque(tid));
        lds_barrier();
        stage_b<-1>(lds, opaque(tid));
        lds_barrier();
        stage_g(lds, opaque(tid));
        lds_barrier();
        residual_window(a, mem, u, lds, opaque(tid));
        lds_barrier();
        stage_a(lds, opaque(tid));
        lds_barrier();
        stage_b<1>(lds, opaque(tid));
        lds_barrier();
        {
          v2f sp[2][8];
#pragma unroll
          for (int h = 0; h < 2; ++h) load_spectrum_half<1>(mem.spec1, 8 * u.c, opaque(tid), h, sp[h]);
          stage_c(lds, lds, twl, opaque(tid));
          wave_sync();
          stage_d(sp, lds, opaque(tid));
        }
        wave_sync();
      }
    } else {
    v2f sp[2][8];
    load_spectrum(mem, u.c, opaque(tid), sp);      // (stage C's arithmetic covers their trip to L2; inside stage D the waves queued up on it)
    stage_c(lds, lds, twl, opaque(tid));
    wave_sync();
    ICS_FFT_STAMP(3);
    stage_d(sp, lds, opaque(tid));
    wave_sync();
    }
    ICS_FFT_STAMP(4);
    load_window(a, mem, decode_unit(a, walk_unit(a, k + G)), opaque(tid), pw, 0, MODE == 0 ? 2 : 1);   // next unit (beyond the last one: dropped accesses); mode 1 holds 64 operand registers through stage G and requests the second tile behind it
    stage_e(lds, lds, twl, opaque(tid));
    v4f fimg[2][4];
    Ops ops;
    if (MODE == 0) load_image(a, mem, u, opaque(tid), fimg);
    else {
      load_ops<TV>(a, mem, u, opaque(tid), 0, ops);
      if (ICS_FFT_M1_EARLY > 0) load_ops<TV>(a, mem, u, opaque(tid), 1, ops, 0, ICS_FFT_M1_EARLY);   // (stages F and G leave registers for part of tile 1's operands)
    }
    lds_barrier();
    ICS_FFT_STAMP(5);
    ICS_FFT_PROBE_COLUMN_PASS(stage_b<-1>(lds, opaque(tid)); lds_barrier(););
    ICS_FFT_STAMP(6);
    stage_g(lds, opaque(tid));
    lds_barrier();
    ICS_FFT_STAMP(7);
    // row-quad epilogue, row group by row group (at most one group's raw values alive beside the operands).  Mode 1 has two operand
    // frames: it requests those of the second tile here and takes its maxima in a second pass, and the second tile of the next unit's
    // window goes out between the passes (registers: 128 per thread with 1024 of them).
    if (MODE >= 1) {
      load_ops<TV>(a, mem, u, opaque(tid), 1, ops, ICS_FFT_M1_EARLY, 4);
      if (ICS_FFT_M1_WIN) load_window(a, mem, decode_unit(a, walk_unit(a, k + G)), opaque(tid), pw, 1, 2);
    }
    Maxima mx; maxima_init(mx);
    v4f res[4][2];
    // lane address and row-group count of the two tiles ONCE per unit (as eight store_quad calls the address arithmetic of the epilogue
    // was 300 of a unit's 1310 vector instructions).  u.ox is wave-uniform: interior tiles -- all but the first and last of a tile row --
    // need no per-pixel column tests.
    QuadOut qo[2];
    const int te = opaque(tid);
#pragma unroll
    for (int t = 0; t < 2; ++t) qo[t].vo = quad_lane(a, u, mem.lout, te, t, qo[t].rows, qo[t].X);
    const bool edge = u.ox[0] < a.ox0 || u.ox[0] + a.V > a.ox1 || u.ox[1] < a.ox0 || u.ox[1] + a.V > a.ox1;
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        read_quads(lds, opaque(tid), i, res[i]);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int e = 0; e < 4; ++e) res[i][t][e] = ICS_FSUB(res[i][t][e], fimg[t][i][e]);        // pyx:488
        asm volatile("" ::: "memory");
      }
      if (edge) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int t = 0; t < 2; ++t) store_quad_at(a, mem, qo[t], true, i, res[i][t]);
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int t = 0; t < 2; ++t) store_quad_at(a, mem, qo[t], false, i, res[i][t]);
      }
    } else {
      // first pass: tile 0 leaves as soon as its maxima are taken (its operands' registers are free for the second window then), tile 1's
      // values wait in res[.][1] for their operands
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        read_quads(lds, opaque(tid), i, res[i]);
        if (edge) { maxima_quad<TV>(a, u, te, 0, i, res[i][0], ops, mx, qo[0], true); store_quad_at(a, mem, qo[0], true, i, res[i][0]); }
        else { maxima_quad<TV>(a, u, te, 0, i, res[i][0], ops, mx, qo[0], false); store_quad_at(a, mem, qo[0], false, i, res[i][0]); }
        asm volatile("" ::: "memory");
      }
      if (!ICS_FFT_M1_WIN) load_window(a, mem, decode_unit(a, walk_unit(a, k + G)), opaque(tid), pw, 1, 2);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (edge) { maxima_quad<TV>(a, u, te, 1, i, res[i][1], ops, mx, qo[1], true); store_quad_at(a, mem, qo[1], true, i, res[i][1]); }
        else { maxima_quad<TV>(a, u, te, 1, i, res[i][1], ops, mx, qo[1], false); store_quad_at(a, mem, qo[1], false, i, res[i][1]); }
      }
    }
    ICS_FFT_STAMP(8);
    if (k + G < a.nunits) {
      store_window(pw, lds, opaque(tid));      // (the slots this thread just read)
      lds_barrier();
      stage_a(lds, opaque(tid));
    }
    ICS_FFT_STAMP(9);
    ICS_FFT_PROBE_TRACE_NEXT();
    if (MODE >= 1) {   // (u.c is uniform)
      uint32_t kg, ku;
      maxima_keys(mx, kg, ku);
#pragma unroll
      for (int c = 0; c < 3; ++c)
        if (u.c == c) { accg[c] = accg[c] > kg ? accg[c] : kg; accu[c] = accu[c] > ku ? accu[c] : ku; }
    }
  }
  if (MODE >= 1) {
    // the workgroup's maxima: wave maxima -> one conditional atomic per wave, channel and value at the END of the kernel (inside the loop
    // the read of the running maximum waited for every store in flight)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const uint32_t kg = ics_wave_max_u32(accg[c]), ku = ics_wave_max_u32(accu[c]);
      if ((tid & 63) == 0) {
        if (kg > a.c.red[ICS_RED_MAXG + c]) atomicMax(a.c.red + ICS_RED_MAXG + c, kg);
        if (ku > a.c.red[ICS_RED_MAXU + c]) atomicMax(a.c.red + ICS_RED_MAXU + c, ku);
      }
    }
  }
}

// ---- A12 + A13 (lib/deconvolution.pyx:567-571): the PSF gradient on the same tiles -----------------------------------------------------------
//     gradk[a, b, c] = sum_{y,x} e'[y, x, c] u[y + pad - a, x + pad - b, c]            (u-frame coordinates; e' = 0 outside the M x N interior)
// Per tile of V x V residual pixels with the 128 x 128 window t of u that starts pad pixels up and left of it:
//     g[a][b] = sum_{v,h<V} e'[v][h] t[v + K-1-a][h + K-1-b] = corr(e' zero-padded, t) at lag (K-1-a, K-1-b) < K  (no wrap-around: v + lag <= 127)
// and corr = IDFT( conj(DFT e') . DFT t ).  The two tiles of a pair travel as real and imaginary part as in the convolutions:
// conj(E0 + i E1) (T0 + i T1) = conj(E0) T0 + conj(E1) T1 + i (...), and the transforms of the first two terms are REAL -- the real part of
// the inverse transform is the sum of both tiles' correlations.  The product is linear: a workgroup keeps ONE channel, adds the products
// of all its tile pairs up in the frequency domain (sixteen complex values per thread) and transforms back once at the end -- two forward
// transforms per pair and no inverse; one K x K block per workgroup, added up in double by k_gradk_fft_reduce in a fixed order.
// fp32 throughout.  Against float64 direct sums on the test frames 1 - 3e-7 of max |gradk| (gate 1e-5); the error scales with
// |e'| |u| of a tile rather than with the sums themselves, so a residual that is pure noise uncorrelated with u is the worst case (4e-5
// estimated for sigma 1e-2 at 600 x 700) -- the matrix-core kernel (ics_gradk_mfma.hip) stays behind conv = ICS_CONV_MATRIX.
template <int DUMMY>
__global__ __launch_bounds__(ICS_FFT_THREADS) void k_gradk_fft(IcsFftArgs a, float* __restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) v2f lds[];
  v2f* const twl = lds + ICS_FFT_P * ICS_FFT_PITCH;
  const int tid = threadIdx.x;
  if (tid < ICS_FFT_TW_ENTRIES) twl[tid] = tw128((tid / ICS_FFT_TWS) * (tid % ICS_FFT_TWS));
  const Mem mem = make_mem(a, 0);                 // in = u, f = e' (the geometry of mode 0: tiles of the M x N interior)
  const int c = (int)blockIdx.x % 3, slot = (int)blockIdx.x / 3, nslots = (int)gridDim.x / 3, npairs = (a.ntiles + 1) / 2;
  v2f acc[2][8];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[s][k] = (v2f){0.f, 0.f};
  for (int p = slot; p < npairs; p += nslots) {
    const Unit u = decode_unit(a, 3 * p + c);
    v4f pe[2][4], pw[2][4];
    load_image(a, mem, u, opaque(tid), pe);       // the residual tiles, zero beyond V x V and beyond the interior
    lds_barrier();                                // (the previous pair's stage D has read the tile)
    store_window(pe, lds, opaque(tid));
    lds_barrier();
    stage_a(lds, opaque(tid));
    load_window(a, mem, u, opaque(tid), pw, 0, 2, a.lag_y, a.lag_x);      // (behind stage A: registers; tap blocks: the window of this launch's lag block)
    lds_barrier();
    stage_b<1>(lds, opaque(tid));
    lds_barrier();
    stage_c<4>(lds, lds, twl, opaque(tid));
    wave_sync();
    v2f ze[2][8];
    stage_d_forward(lds, opaque(tid), ze);
    lds_barrier();
    store_window(pw, lds, opaque(tid));
    lds_barrier();
    stage_a(lds, opaque(tid));
    lds_barrier();
    stage_b<1>(lds, opaque(tid));
    lds_barrier();
    stage_c<4>(lds, lds, twl, opaque(tid));
    wave_sync();
    v2f zu[2][8];
    stage_d_forward(lds, opaque(tid), zu);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[s][k] += cmulc(zu[s][k], ze[s][k]);     // += DFT(t) conj(DFT(e'))
  }
  lds_barrier();
  stage_d_inverse(acc, lds, opaque(tid));
  wave_sync();
  stage_e(lds, lds, twl, opaque(tid));
  lds_barrier();
  stage_b<-1>(lds, opaque(tid));
  lds_barrier();
  stage_g(lds, opaque(tid));
  lds_barrier();
  const int K = a.c.g.K;
  if (a.blk_k) {   // tap blocks: the blk_k x blk_k lags of this launch's block, in lag order (k_gradk_fft_reduce_blk places them)
    const int Kb = a.blk_k;
    for (int i = tid; i < Kb * Kb; i += ICS_FFT_THREADS) {
      const int ly = i / Kb, lx = i - ly * Kb;
      partial[(size_t)blockIdx.x * Kb * Kb + i] = lds[ly * ICS_FFT_PITCH + lx].x * (1.0f / (ICS_FFT_P * ICS_FFT_P));
    }
    return;
  }
  for (int i = tid; i < K * K; i += ICS_FFT_THREADS) {
    const int aa = i / K, bb = i - aa * K;
    partial[(size_t)blockIdx.x * K * K + i] = lds[(K - 1 - aa) * ICS_FFT_PITCH + (K - 1 - bb)].x * (1.0f / (ICS_FFT_P * ICS_FFT_P));
  }
}

// ---- PSF sizes above ICS_FFT_MAX_K: tap blocks on the tiles -----------------------------------------------------------------------------------
// A tile keeps 128 - K + 1 of its 128 pixels a side: 32 at 97, nothing at 129.  A convolution is linear in its taps, so the K x K PSF is cut
// into blk_n x blk_n blocks of blk_k x blk_k taps (blk_k <= 65) and block (qa, qb) is the blk_k x blk_k kernel on the window that starts
// (qa blk_k, qb blk_k) further down / right:
//     out = sum_q IDFT( S_q . DFT(window_q) ) = IDFT( sum_q S_q . DFT(window_q) )
// -- blk_n^2 forward transforms whose products meet in the frequency domain (sixteen complex values per thread), ONE inverse transform and
// one epilogue per unit, with tiles of 128 - blk_k + 1 valid pixels a side.  The epilogues are those of modes 0 and 1 (shipped loop).
template <int MODE>
__global__ __launch_bounds__(ICS_FFT_THREADS) void k_conv_fft_blk(IcsFftArgs a) {
  extern __shared__ __attribute__((aligned(16))) v2f lds[];
  v2f* const twl = lds + ICS_FFT_P * ICS_FFT_PITCH;
  const int tid = threadIdx.x;
  const int G = gridDim.x;
  if (tid < ICS_FFT_TW_ENTRIES) twl[tid] = tw128((tid / ICS_FFT_TWS) * (tid % ICS_FFT_TWS));
  const Mem mem = make_mem(a, MODE);
  const int q = (G & 7) ? (int)blockIdx.x : (int)(blockIdx.x & 7) * (G >> 3) + (int)(blockIdx.x >> 3);
  uint32_t accg[3] = {0u, 0u, 0u}, accu[3] = {0u, 0u, 0u};
  const int nblk = a.blk_n * a.blk_n;
  // (the blocks' windows are NOT requested a block ahead: the running sum, the block's spectrum and a window in flight through the runtime
  //  loop over the blocks spill 40 registers; every block waits out its window's round trip -- measured 5-10 x ahead of the matrix cores' tap
  //  blocks as it is)
  for (int n = q; n < a.nunits; n += G) {
    const Unit u = decode_unit(a, n);
    v2f acc[2][8];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[s][k] = (v2f){0.f, 0.f};
    for (int b = 0; b < nblk; ++b) {
      const int qa = b / a.blk_n, qb = b - qa * a.blk_n;
      {
        v4f pw[2][4];
        load_window(a, mem, u, opaque(tid), pw, 0, 2, qa * a.blk_k, qb * a.blk_k);
        lds_barrier();                            // (the previous block's stage D / the previous unit's epilogue has read the tile)
        store_window(pw, lds, opaque(tid));
      }
      lds_barrier();
      stage_a(lds, opaque(tid));
      lds_barrier();
      stage_b<1>(lds, opaque(tid));
      lds_barrier();
      v2f sp[2][8];
#pragma unroll
      for (int h = 0; h < 2; ++h) load_spectrum_half<1>(mem.spec, 8 * (3 * b + u.c), opaque(tid), h, sp[h]);
      stage_c<4>(lds, lds, twl, opaque(tid));
      wave_sync();
      stage_d_mac(lds, opaque(tid), sp, acc);
    }
    lds_barrier();
    stage_d_inverse(acc, lds, opaque(tid));
    wave_sync();
    stage_e(lds, lds, twl, opaque(tid));
    lds_barrier();
    stage_b<-1>(lds, opaque(tid));
    lds_barrier();
    stage_g(lds, opaque(tid));
    QuadOut qo[2];
    const int te = opaque(tid);
#pragma unroll
    for (int t = 0; t < 2; ++t) qo[t].vo = quad_lane(a, u, mem.lout, te, t, qo[t].rows, qo[t].X);
    const bool edge = u.ox[0] < a.ox0 || u.ox[0] + a.V > a.ox1 || u.ox[1] < a.ox0 || u.ox[1] + a.V > a.ox1;
    if (MODE == 0) {
      v4f fimg[2][4];
      load_image(a, mem, u, opaque(tid), fimg);
      lds_barrier();
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        v4f r[2];
        read_quads(lds, te, i, r);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
          for (int e = 0; e < 4; ++e) r[t][e] = ICS_FSUB(r[t][e], fimg[t][i][e]);        // pyx:488
          store_quad_at(a, mem, qo[t], edge, i, r[t]);
        }
      }
    } else {
      Ops ops;
      load_ops<false>(a, mem, u, opaque(tid), 0, ops);
      load_ops<false>(a, mem, u, opaque(tid), 1, ops);
      lds_barrier();
      Maxima mx; maxima_init(mx);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        v4f r[2];
        read_quads(lds, te, i, r);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          maxima_quad<false>(a, u, te, t, i, r[t], ops, mx, qo[t], edge);
          store_quad_at(a, mem, qo[t], edge, i, r[t]);
        }
      }
      uint32_t kg, ku;
      maxima_keys(mx, kg, ku);
#pragma unroll
      for (int c = 0; c < 3; ++c)
        if (u.c == c) { accg[c] = accg[c] > kg ? accg[c] : kg; accu[c] = accu[c] > ku ? accu[c] : ku; }
    }
  }
  if (MODE == 1) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const uint32_t kg = ics_wave_max_u32(accg[c]), ku = ics_wave_max_u32(accu[c]);
      if ((tid & 63) == 0) {
        if (kg > a.c.red[ICS_RED_MAXG + c]) atomicMax(a.c.red + ICS_RED_MAXG + c, kg);
        if (ku > a.c.red[ICS_RED_MAXU + c]) atomicMax(a.c.red + ICS_RED_MAXU + c, ku);
      }
    }
  }
}

// ---- the image side of mode 2: F = DFT of the two image windows of every unit (unnormalised), once per image ----------------------------------
// k_conv_fft<2> subtracts it from 16384 S0 T in stage D -- the "- image" of pyx:488 in the frequency domain.  The window of a unit's tile
// starts (pad, pad) before the tile's first output pixel (where the residual window of the back-projection starts); the image frame is zero
// outside the M x N interior.  Stored in load_spectrum's order, one block of 8 x 1024 quads per unit: 128 KB per unit, read once per inner
// iteration (HBM), written once per upload of the image.  `a` = the mode-2 geometry with in = the image's planar mirror and wpad = pad.
template <int DUMMY>
__global__ __launch_bounds__(ICS_FFT_THREADS) void k_fft_image_spectrum(IcsFftArgs a) {
  extern __shared__ __attribute__((aligned(16))) v2f lds[];
  v2f* const twl = lds + ICS_FFT_P * ICS_FFT_PITCH;
  const int tid = threadIdx.x;
  if (tid < ICS_FFT_TW_ENTRIES) twl[tid] = tw128((tid / ICS_FFT_TWS) * (tid % ICS_FFT_TWS));
  const Mem mem = make_mem(a, 2);
  for (int n = blockIdx.x; n < a.nunits; n += gridDim.x) {
    const Unit u = decode_unit(a, n);
    v4f pw[2][4];
    load_window(a, mem, u, opaque(tid), pw);
    lds_barrier();                                // (the previous unit's stage D has read the tile)
    store_window(pw, lds, opaque(tid));
    lds_barrier();
    stage_a(lds, opaque(tid));
    lds_barrier();
    stage_b<1>(lds, opaque(tid));
    lds_barrier();
    stage_c(lds, lds, twl, opaque(tid));
    wave_sync();
    v2f z[2][8];
    stage_d_forward(lds, opaque(tid), z);
    store_spectrum(mem.fspec, 8 * n, opaque(tid), z);
  }
}

// ---- A11 + A12 + A13 (pyx:555-571) as ONE unit on the tiles: three transforms where k_conv_fft<0> + k_gradk_fft run four -------------------
// Per tile pair and channel, with t = the two 128 x 128 windows of u (real / imaginary part):
//     T = DFT(t)                                          A B C D        kept in registers (sixteen values per thread)
//     r = IDFT(S T);  e' = (r - image) on the valid V x Vy pixels inside the interior, 0 elsewhere       D E F G + row-quad epilogue, IN the tile buffer
//     acc += T conj(DFT(e'))                              A B C D        the workgroup's running sum, as k_gradk_fft
// The residual never leaves the CU (it is stored only under the stop-test window, whose statistics read it: pyx:600-601, 627), the window
// is read once instead of twice and transformed once.  The same stage functions in the same order as the two kernels it replaces and the
// same walk (workgroup = channel blockIdx % 3, pairs slot, slot + nslots, ...): e' and the K x K blocks are bit-identical to theirs.
// Registers (1024 threads: 128): acc and T stay alive through the unit, so the sixteen-point stages run in their lean forms and the
// two prefetches sit beside eight-point stages only: the image quads are requested behind stage G's last LDS write (in flight through the
// barrier), the next unit's window in front of the second stage D.
#ifndef ICS_FFT_FUSED_IMG_EARLY
#define ICS_FFT_FUSED_IMG_EARLY 2
#endif
template <int DUMMY>
__global__ __launch_bounds__(ICS_FFT_THREADS) void k_synth_gradk_fft(IcsFftArgs a, float* __restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) v2f lds[];
  v2f* const twl = lds + ICS_FFT_P * ICS_FFT_PITCH;
  const int tid = threadIdx.x;
  if (tid < ICS_FFT_TW_ENTRIES) twl[tid] = tw128((tid / ICS_FFT_TWS) * (tid % ICS_FFT_TWS));
  const Mem mem = make_mem(a, 0);                 // in = u, f = image, out = e' (the geometry of mode 0: tiles of the M x N interior)
  const int c = (int)blockIdx.x % 3, slot = (int)blockIdx.x / 3, nslots = (int)gridDim.x / 3, npairs = (a.ntiles + 1) / 2;
  v2f acc[2][8];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[s][k] = (v2f){0.f, 0.f};
  if (slot < npairs) {
    v4f pw[2][4];
    load_window(a, mem, decode_unit(a, 3 * slot + c), opaque(tid), pw);
    store_window(pw, lds, opaque(tid));
    lds_barrier();
    stage_a(lds, opaque(tid));
  }
  for (int p = slot; p < npairs; p += nslots) {
    const Unit u = decode_unit(a, 3 * p + c);
    lds_barrier();
    stage_b<1>(lds, opaque(tid));
    lds_barrier();
    v2f zu[2][8];
    {
      v2f sp[2][8];
      load_spectrum(mem, c, opaque(tid), sp);
      stage_c<4>(lds, lds, twl, opaque(tid));
      wave_sync();
      stage_d_keep(sp, lds, opaque(tid), zu);
    }
    wave_sync();
    stage_e_lean(lds, lds, twl, opaque(tid));
    lds_barrier();
    stage_b<-1>(lds, opaque(tid));
    lds_barrier();
    stage_g(lds, opaque(tid));
    {
      // the image quads in two halves of two row groups: the first is requested behind stage G's last LDS write (in flight through the
      // barrier), the second in front of the first half's arithmetic -- 96 registers of spectra and image beside the epilogue otherwise
      v4f fimg[2][4];
      load_image_rows(a, mem, u, opaque(tid), fimg, 0, ICS_FFT_FUSED_IMG_EARLY);
      lds_barrier();
      load_image_rows(a, mem, u, opaque(tid), fimg, ICS_FFT_FUSED_IMG_EARLY, 4);
      QuadOut qo[2];
      const int te = opaque(tid);
#pragma unroll
      for (int t = 0; t < 2; ++t) qo[t].vo = quad_lane(a, u, mem.lout, te, t, qo[t].rows, qo[t].X);
      const bool edge = u.ox[0] < a.ox0 || u.ox[0] + a.V > a.ox1 || u.ox[1] < a.ox0 || u.ox[1] + a.V > a.ox1;
      bool store = a.store_all != 0;
#pragma unroll
      for (int t = 0; t < 2; ++t) store = store || (u.has[t] && u.oy[t] < a.wy1 && u.oy[t] + a.Vy > a.wy0 && u.ox[t] < a.wx1 && u.ox[t] + a.V > a.wx0);   // (uniform)
      if (store) {
#pragma unroll
        for (int i = 0; i < 4; ++i) residual_quads(a, mem, qo, edge, true, lds, te, i, fimg);
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) residual_quads(a, mem, qo, edge, false, lds, te, i, fimg);
      }
    }
    lds_barrier();
    stage_a(lds, opaque(tid));
    lds_barrier();
    stage_b<1>(lds, opaque(tid));
    lds_barrier();
    stage_c<4>(lds, lds, twl, opaque(tid));
    wave_sync();
    {
      v4f pw[2][4];
      load_window(a, mem, decode_unit(a, 3 * (p + nslots) + c), opaque(tid), pw);   // next unit (beyond the last one: dropped accesses)
      stage_d_acc(lds, opaque(tid), zu, acc);
      if (p + nslots < npairs) {
        lds_barrier();                              // (every wave has read its rows)
        store_window(pw, lds, opaque(tid));
        lds_barrier();
        stage_a(lds, opaque(tid));
      }
    }
  }
  lds_barrier();
  stage_d_inverse(acc, lds, opaque(tid));
  wave_sync();
  stage_e(lds, lds, twl, opaque(tid));
  lds_barrier();
  stage_b<-1>(lds, opaque(tid));
  lds_barrier();
  stage_g(lds, opaque(tid));
  lds_barrier();
  const int K = a.c.g.K;
  for (int i = tid; i < K * K; i += ICS_FFT_THREADS) {
    const int aa = i / K, bb = i - aa * K;
    partial[(size_t)blockIdx.x * K * K + i] = lds[(K - 1 - aa) * ICS_FFT_PITCH + (K - 1 - bb)].x * (1.0f / (ICS_FFT_P * ICS_FFT_P));
  }
}

// tap blocks: lag (lag_y + ly, lag_x + lx) is tap (K - 1 - lag_y - ly, K - 1 - lag_x - lx) of the gradient; one wave per value as below
__global__ __launch_bounds__(256) void k_gradk_fft_reduce_blk(const float* __restrict__ partial, int nblocks, int K, int Kb, int lag_y, int lag_x, float* __restrict__ gradk) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= 3 * Kb * Kb) return;
  const int c = i % 3, l = i / 3, ly = l / Kb, lx = l - ly * Kb;
  const int aa = K - 1 - lag_y - ly, bb = K - 1 - lag_x - lx;
  if (aa < 0 || bb < 0) return;
  double s = 0.0;
  for (int b = c + 3 * lane; b < nblocks; b += 192) s += (double)partial[(size_t)b * Kb * Kb + l];
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
  if (lane == 0) gradk[(aa * K + bb) * 3 + c] = (float)s;
}

// gradk[a][b][c] = sum of the blocks of the workgroups that kept channel c (block % 3 == c), in double.  One wave per value: lane l adds
// blocks c + 3 l, c + 3 (l + 64), ... and the 64 lane sums meet in a fixed butterfly (the same bits run after run).  (One thread per value
// with its 85 serial loads took 27 us, 7 % of the gradient kernel it follows.)
__global__ __launch_bounds__(256) void k_gradk_fft_reduce(const float* __restrict__ partial, int nblocks, int K, float* __restrict__ gradk) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= 3 * K * K) return;
  const int c = i % 3, ab = i / 3;
  double s = 0.0;
  for (int b = c + 3 * lane; b < nblocks; b += 192) s += (double)partial[(size_t)b * K * K + ab];
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
  if (lane == 0) gradk[i] = (float)s;
}

// ---- spectrum: S_o,c[ky][kx] = conj( sum_{a,b} W_o[a][b][c] w^(a ky + b kx) ) / 128^2,  w = exp(-2 pi i / 128), stored at spec_index(c, ky, kx) ----
// W_0 = rot180(psf) (mode 0), W_1 = psf (mode 1).  Double accumulation (a PSF value enters with its float32 value, the twiddles from a
// double table built on the device); one workgroup per (orientation, channel, 32 columns kx): G[a][kx] = sum_b W[a][b] w^(b kx) in LDS,
// then S[ky][kx] = conj(sum_a G[a][kx] w^(a ky)).
// (tap blocks: blockIdx.y = block q = qa * nb + qb of Kb x Kb taps starting at W_o[qa Kb][qb Kb], taps beyond K are zero; its spectra go to
//  spec + q * 3 * 128 * 128.  nb = 1, Kb = K: the whole PSF.)
__global__ __launch_bounds__(256) void k_fft_spectrum(const float* __restrict__ psf, int K, v2f* __restrict__ spec0, v2f* __restrict__ spec1, int nb, int Kb) {
  extern __shared__ __attribute__((aligned(16))) double sm[];   // [128][2] twiddles, then [K][32][2] G
  double* twd = sm;
  double* Gs = sm + 256;
  const int tid = threadIdx.x;
  // one workgroup per (orientation, channel, 32 columns kx, 32 rows ky): 96 of them; each builds the G of its columns itself (K^2 x 32
  // products) and then its 32 x 32 values of S (K each).  (24 workgroups with all 128 rows each took 30 us at 31 x 31, 65 us at 63 x 63 --
  // per inner iteration of a blind run.)
  const int o = blockIdx.x / 48, c = (blockIdx.x / 16) % 3, kx0 = ((blockIdx.x >> 2) & 3) * 32, ky0 = (blockIdx.x & 3) * 32;
  const int qa = (int)blockIdx.y / nb, qb = (int)blockIdx.y - qa * nb, a0 = qa * Kb, b0 = qb * Kb;
  spec0 += (size_t)blockIdx.y * 3 * ICS_FFT_P * ICS_FFT_P; spec1 += (size_t)blockIdx.y * 3 * ICS_FFT_P * ICS_FFT_P;
  if (tid < 128) {
    double sn, cs;
    sincospi((double)tid / 64.0, &sn, &cs);
    twd[2 * tid] = cs; twd[2 * tid + 1] = -sn;
  }
  __syncthreads();
  for (int i = tid; i < Kb * 32; i += 256) {
    const int aa = i >> 5, kx = kx0 + (i & 31);
    double re = 0.0, im = 0.0;
    for (int b = 0; b < Kb; ++b) {
      const int ta = a0 + aa, tb = b0 + b;           // the tap of W_o this is
      double wv = 0.0;
      if (ta < K && tb < K) wv = o == 0 ? (double)psf[((K - 1 - ta) * K + (K - 1 - tb)) * 3 + c] : (double)psf[(ta * K + tb) * 3 + c];
      const int t = (b * kx) & 127;
      re += wv * twd[2 * t]; im += wv * twd[2 * t + 1];
    }
    Gs[2 * i] = re; Gs[2 * i + 1] = im;
  }
  __syncthreads();
  for (int i = tid; i < 32 * 32; i += 256) {
    const int ky = ky0 + (i >> 5), kxl = i & 31;
    double re = 0.0, im = 0.0;
    for (int aa = 0; aa < Kb; ++aa) {
      const double gr = Gs[2 * (aa * 32 + kxl)], gi = Gs[2 * (aa * 32 + kxl) + 1];
      const int t = (aa * ky) & 127;
      const double wr = twd[2 * t], wi = twd[2 * t + 1];
      re += gr * wr - gi * wi; im += gr * wi + gi * wr;
    }
    const double sc = 1.0 / (128.0 * 128.0);
    (o == 0 ? spec0 : spec1)[spec_index(c, ky, kx0 + kxl)] = (v2f){(float)(re * sc), (float)(-im * sc)};
  }
}

}  // namespace icsfft

// ---- launchers -----------------------------------------------------------------------------------------------------------------------------
// (the stage functions are exact for any K <= 125; what bounds the range is the valid part of a tile, 128 - K + 1 pixels a side: 44 at 85)
bool ics_conv_fft_supported(int K) { return K >= 3 && K <= ICS_FFT_MAX_K && (K & 1); }
size_t ics_conv_fft_spectrum_floats() { return (size_t)3 * ICS_FFT_P * ICS_FFT_P * 2; }   // per orientation (and per tap block)
// tap blocks for PSF sizes above ICS_FFT_MAX_K: the fewest blocks per axis whose size stays at 65 or below (2 to 129, 3 to 193, 4 to 255)
bool ics_conv_fft_blk_supported(int K) { return K > ICS_FFT_MAX_K && K <= 255 && (K & 1); }
void ics_conv_fft_blk_shape(int K, int* blk_n, int* blk_k) {
  int n = (K + 64) / 65;
  int k = (K + n - 1) / n;
  *blk_n = n; *blk_k = k;
}

hipError_t ics_launch_fft_spectrum(const float* psf, int K, float* spec_conv, float* spec_corr, hipStream_t s, int blk_n, int blk_k) {
  const int nb = blk_n > 0 ? blk_n : 1, Kb = blk_n > 0 ? blk_k : K;
  const size_t lds = (256 + (size_t)Kb * 32 * 2) * sizeof(double);   // 35 KB at 65, 52 KB at 97
  hipLaunchKernelGGL(icsfft::k_fft_spectrum, dim3(96, nb * nb), dim3(256), lds, s, psf, K, reinterpret_cast<v2f*>(spec_conv), reinterpret_cast<v2f*>(spec_corr), nb, Kb);
  return hipGetLastError();
}

void ics_conv_fft_fill_args(int mode, const IcsConvArgs& c, const float* spec, IcsFftArgs* a, int blk_n = 0, int blk_k = 0) {
  a->c = c;
  a->trace = nullptr;
  a->planar = 0;
  a->wy0 = a->wy1 = a->wx0 = a->wx1 = 0; a->store_all = 0;
  a->wpad = c.g.pad; a->fspec = nullptr; a->spec1 = nullptr; a->lag_y = a->lag_x = 0; a->rot = 0;
  a->spec = reinterpret_cast<const v2f*>(spec);
  const IcsGeom& g = c.g;
  a->blk_n = blk_n; a->blk_k = blk_k;
  a->Vy = ICS_FFT_P - (blk_k ? blk_k : g.K) + 1;   // valid rows per tile: all of them (tap blocks: of the block's size)
  a->V = a->Vy & ~3;                     // valid pixels per tile row, whole quads (16-byte stores never straddle two tiles)
  if (mode == 2) { a->Vy = ICS_FFT_P - 2 * g.K + 2; a->V = a->Vy & ~3; a->wpad = 2 * g.pad; }   // A1 + A3 in one unit: the valid part of two convolutions in a row
  if (mode == 0) { a->oy0 = g.pad; a->ox0 = g.pad; a->oy1 = g.pad + g.M; a->ox1 = g.pad + g.N; }
  else { a->oy0 = 0; a->ox0 = 0; a->oy1 = g.uM; a->ox1 = g.uN; }
  a->gx0 = a->ox0 & ~3;
  a->tiles_x = (a->ox1 - a->gx0 + a->V - 1) / a->V;
  const int tiles_y = (a->oy1 - a->oy0 + a->Vy - 1) / a->Vy;
  a->ntiles = a->tiles_x * tiles_y;
  a->tiles_x_magic = 0x100000000ull / (unsigned)a->tiles_x + 1ull;
  a->nunits = 3 * ((a->ntiles + 1) / 2);
}

hipError_t ics_launch_conv_fft_args(int mode, const IcsFftArgs& a, hipStream_t s);
hipError_t ics_launch_conv_fft_region(const IcsConvArgs& c, const float* spec, int oy0, int ox0, int oy1, int ox1, hipStream_t s);
hipError_t ics_launch_conv_fft(int mode, const IcsConvArgs& c, const float* spec, int planar, hipStream_t s) {
  if (mode != 0 && mode != 1) return hipErrorInvalidValue;   // (mode 2: ics_launch_conv2_fft)
  if (planar != ICS_FFT_PL_ALL) return hipErrorInvalidValue;   // every frame a channel-planar mirror: the kernel moves 4 pixels of a plane row per access
  IcsFftArgs a;
  ics_conv_fft_fill_args(mode, c, spec, &a);
  a.planar = planar;
  return ics_launch_conv_fft_args(mode, a, s);
}
// mode 0 over a part of the interior only: the tiles that cover output rows [oy0, oy1) x columns [ox0, ox1) of the u-frame (a window of the
// residual; what lies outside the region inside a stored quad is written as zero, the rest of the frame is not touched)
hipError_t ics_launch_conv_fft_region(const IcsConvArgs& c, const float* spec, int oy0, int ox0, int oy1, int ox1, hipStream_t s) {
  IcsFftArgs a;
  ics_conv_fft_fill_args(0, c, spec, &a);
  a.planar = ICS_FFT_PL_ALL;
  a.oy0 = oy0; a.ox0 = ox0; a.oy1 = oy1; a.ox1 = ox1;
  a.gx0 = a.ox0 & ~3;
  a.tiles_x = (a.ox1 - a.gx0 + a.V - 1) / a.V;
  const int tiles_y = (a.oy1 - a.oy0 + a.Vy - 1) / a.Vy;
  a.ntiles = a.tiles_x * tiles_y;
  a.tiles_x_magic = 0x100000000ull / (unsigned)a.tiles_x + 1ull;
  a.nunits = 3 * ((a.ntiles + 1) / 2);
  return ics_launch_conv_fft_args(0, a, s);
}
hipError_t ics_launch_conv_fft_args(int mode, const IcsFftArgs& a, hipStream_t s) {
  static std::atomic<bool> configured[5][ICS_MAX_DEVICES];
  const int dev = ics_current_device();
  int grid = ics_device_cus(dev);
  if (const int mw = ics_debug().max_wgs.load(std::memory_order_relaxed); mw > 0 && grid > mw) grid = mw;
  if (grid > a.nunits) grid = a.nunits;
  // (mode 1 with the T frame: the PAM kinds, whose epilogue takes u and T where the shipped loop takes u and ut.  The active MM-TV kind
  //  needs all three and does not fit 128 registers: not built)
  if (a.c.tv && a.c.tv_kind == 1) return hipErrorInvalidValue;
  const bool pam = mode == 1 && a.c.tv && a.c.tv_kind >= 2;
  auto k0 = icsfft::k_conv_fft<0, false>;
  auto k1 = icsfft::k_conv_fft<1, false>;
  auto k1t = icsfft::k_conv_fft<1, true>;
  auto k2 = icsfft::k_conv_fft<2, false>;
  auto k2t = icsfft::k_conv_fft<2, true>;
  const bool pam2 = mode == 2 && a.c.tv && a.c.tv_kind >= 2;      // (the PAM kinds' epilogue on mode 2's units: operands u and T, G = T + lambd gradu stored)
  if (mode == 2 && (!a.spec1 || !a.fspec)) return hipErrorInvalidValue;
  auto kern = mode == 2 ? (pam2 ? k2t : k2) : (mode == 0 ? k0 : (pam ? k1t : k1));
  const int slot = mode == 2 ? (pam2 ? 4 : 3) : (pam ? 2 : mode);
  if (hipError_t e = ics_configure_lds(configured[slot], dev, kern, ICS_FFT_LDS_BYTES); e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(ICS_FFT_THREADS), ICS_FFT_LDS_BYTES, s, a);
  return hipGetLastError();
}
// A12 + A13 on the transform tiles: u and e = origins of channel-planar mirrors; partial: ics_gradk_fft_blocks() * K * K floats
int ics_gradk_fft_blocks(int cus) {
  int grid = (cus / 3) * 3;
  if (const int mw = ics_debug().max_wgs.load(std::memory_order_relaxed); mw > 0 && grid > mw) grid = (mw / 3) * 3;
  return grid < 3 ? 3 : grid;
}
hipError_t ics_launch_gradk_fft(const float* u, const float* e, const IcsGeom& g, float* partial, float* gradk, hipStream_t s) {
  IcsConvArgs c;
  memset(&c, 0, sizeof c);
  c.g = g; c.in = u; c.f = e; c.out = const_cast<float*>(e); c.u = u; c.ut = u;
  IcsFftArgs a;
  ics_conv_fft_fill_args(0, c, nullptr, &a);
  a.planar = ICS_FFT_PL_ALL;
  static std::atomic<bool> configured[ICS_MAX_DEVICES];
  const int dev = ics_current_device();
  int grid = ics_gradk_fft_blocks(ics_device_cus(dev));
  const int npairs = (a.ntiles + 1) / 2;
  if (grid > 3 * npairs) grid = 3 * npairs;
  auto kern = icsfft::k_gradk_fft<0>;
  if (hipError_t err = ics_configure_lds(configured, dev, kern, ICS_FFT_LDS_BYTES); err != hipSuccess) return err;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(ICS_FFT_THREADS), ICS_FFT_LDS_BYTES, s, a, partial);
  hipLaunchKernelGGL(icsfft::k_gradk_fft_reduce, dim3((3 * g.K * g.K + 3) / 4), dim3(256), 0, s, partial, grid, g.K, gradk);
  return hipGetLastError();
}
// A11 + A12 + A13 in one kernel: u, f, e = origins of channel-planar mirrors; spec = the convolution orientation's spectrum; the window
// (u-frame coordinates) says which tiles store their residual; partial: ics_gradk_fft_blocks() * K * K floats
hipError_t ics_launch_synth_gradk_fft(const float* u, const float* f, float* e, const float* spec, const IcsGeom& g, int wy0, int wy1, int wx0, int wx1, int store_all,
                                      float* partial, float* gradk, hipStream_t s) {
  IcsConvArgs c;
  memset(&c, 0, sizeof c);
  c.g = g; c.in = u; c.f = f; c.out = e; c.u = u; c.ut = u;
  IcsFftArgs a;
  ics_conv_fft_fill_args(0, c, spec, &a);
  a.planar = ICS_FFT_PL_ALL;
  a.wy0 = wy0; a.wy1 = wy1; a.wx0 = wx0; a.wx1 = wx1; a.store_all = store_all;
  static std::atomic<bool> configured[ICS_MAX_DEVICES];
  const int dev = ics_current_device();
  int grid = ics_gradk_fft_blocks(ics_device_cus(dev));
  const int npairs = (a.ntiles + 1) / 2;
  if (grid > 3 * npairs) grid = 3 * npairs;
  auto kern = icsfft::k_synth_gradk_fft<0>;
  if (hipError_t err = ics_configure_lds(configured, dev, kern, ICS_FFT_LDS_BYTES); err != hipSuccess) return err;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(ICS_FFT_THREADS), ICS_FFT_LDS_BYTES, s, a, partial);
  hipLaunchKernelGGL(icsfft::k_gradk_fft_reduce, dim3((3 * g.K * g.K + 3) / 4), dim3(256), 0, s, partial, grid, g.K, gradk);
  return hipGetLastError();
}
// ---- mode 2: A1 + A2 + A3 in one unit per tile pair (k_conv_fft<2>) ---------------------------------------------------------------------------
// valid output per tile: 128 - 2 K + 2 pixels a side; the per-unit image spectra must stay addressable with 32-bit byte offsets
size_t ics_conv2_fft_fspec_floats(const IcsGeom& g) {
  IcsConvArgs c; memset(&c, 0, sizeof c); c.g = g;
  IcsFftArgs a;
  ics_conv_fft_fill_args(2, c, nullptr, &a);
  return (size_t)a.nunits * 8 * ICS_FFT_THREADS * 4;
}
bool ics_conv2_fft_supported(const IcsGeom& g) {
  if (!ics_conv_fft_supported(g.K) || ICS_FFT_P - 2 * g.K + 2 < 16) return false;
  return ics_conv2_fft_fspec_floats(g) * sizeof(float) < 0x7FFFFFFFull;
}
// f = origin of the image's planar mirror; fspec = ics_conv2_fft_fspec_floats(g) floats
hipError_t ics_launch_fft_image_spectrum(const float* f, const IcsGeom& g, float* fspec, hipStream_t s) {
  IcsConvArgs c;
  memset(&c, 0, sizeof c);
  c.g = g; c.in = f; c.f = f; c.out = const_cast<float*>(f); c.u = f; c.ut = f;
  IcsFftArgs a;
  ics_conv_fft_fill_args(2, c, nullptr, &a);
  a.planar = ICS_FFT_PL_ALL; a.wpad = g.pad; a.fspec = fspec; a.spec = reinterpret_cast<const v2f*>(fspec); a.spec1 = a.spec;
  static std::atomic<bool> configured[ICS_MAX_DEVICES];
  const int dev = ics_current_device();
  int grid = ics_device_cus(dev);
  if (const int mw = ics_debug().max_wgs.load(std::memory_order_relaxed); mw > 0 && grid > mw) grid = mw;
  if (grid > a.nunits) grid = a.nunits;
  auto kern = icsfft::k_fft_image_spectrum<0>;
  if (hipError_t err = ics_configure_lds(configured, dev, kern, ICS_FFT_LDS_BYTES); err != hipSuccess) return err;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(ICS_FFT_THREADS), ICS_FFT_LDS_BYTES, s, a);
  return hipGetLastError();
}
// c: in = u = the u mirror's origin, out = the back-projection's, f = the image's, ut, red, lambd as for mode 1; spec_conv / spec_corr = the two
// weight spectra; fspec = the image spectra of THIS image and geometry
hipError_t ics_launch_conv2_fft(const IcsConvArgs& c, const float* spec_conv, const float* spec_corr, const float* fspec, hipStream_t s) {
  IcsFftArgs a;
  ics_conv_fft_fill_args(2, c, spec_conv, &a);
  a.planar = ICS_FFT_PL_ALL;
  a.spec1 = reinterpret_cast<const v2f*>(spec_corr); a.fspec = const_cast<float*>(fspec);
  // Where the static walk starts: the units of the outer ring take four transforms instead of two (about 1.6 of a unit's time), and a walk from
  // the first tile row ends on the last one -- the final, partial round of units is then made of the heaviest units (4096^2 / 15: 86 units of
  // the bottom row on 86 workgroups while 170 idle: 0.293 -> 0.281 ms with the walk started half way).  Of eight starting points the one with
  // the lightest most-loaded workgroup is taken (workgroup of walk position k = k mod grid); the choice depends on the geometry and the grid
  // only and is kept for the next launch.  Order only: results do not change.
  if (ics_debug().fft_rot.load(std::memory_order_relaxed)) {
    static std::atomic<long long> cache_key{-1};
    static std::atomic<int> cache_rot{0};
    const int dev = ics_current_device();
    int grid = ics_device_cus(dev);
    if (const int mw = ics_debug().max_wgs.load(std::memory_order_relaxed); mw > 0 && grid > mw) grid = mw;
    if (grid > a.nunits) grid = a.nunits;
    const long long key = ((long long)c.g.M << 40) ^ ((long long)c.g.N << 16) ^ ((long long)c.g.K << 8) ^ (long long)grid;
    if (cache_key.load(std::memory_order_acquire) == key) a.rot = cache_rot.load(std::memory_order_relaxed);
    else {
      std::vector<unsigned char> ring((size_t)a.nunits);
      for (int n = 0; n < a.nunits; ++n) ring[n] = icsfft::unit_is_border(a, icsfft::decode_unit(a, n)) ? 1 : 0;
      std::vector<int> load((size_t)grid);
      int best = 0; long best_cost = -1;
      static const int order[8] = {4, 0, 2, 6, 1, 3, 5, 7};      // (ties: the walk started half way first -- the form that was measured)
      for (int ci = 0; ci < 8; ++ci) {
        const int cand = order[ci];
        a.rot = (int)((long long)a.nunits * cand / 8);
        std::fill(load.begin(), load.end(), 0);
        for (int k = 0; k < a.nunits; ++k) load[k % grid] += ring[icsfft::walk_unit(a, k)] ? 16 : 10;      // (tenths of a two-transform unit)
        const long cost = *std::max_element(load.begin(), load.end());
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = a.rot; }
      }
      a.rot = best;
      cache_rot.store(best, std::memory_order_relaxed); cache_key.store(key, std::memory_order_release);
    }
  }
  return ics_launch_conv_fft_args(2, a, s);
}
// ---- tap blocks (PSF sizes above ICS_FFT_MAX_K) ------------------------------------------------------------------------------------------------
// spec = this orientation's block spectra, blk_n^2 x [3][128][128] (ics_launch_fft_spectrum with blk_n, blk_k)
hipError_t ics_launch_conv_fft_blk(int mode, const IcsConvArgs& c, const float* spec, int blk_n, int blk_k, hipStream_t s) {
  if ((mode != 0 && mode != 1) || blk_n < 2 || blk_k < 3 || blk_k > 65 || (c.tv && c.tv_kind)) return hipErrorInvalidValue;   // (shipped loop)
  IcsFftArgs a;
  ics_conv_fft_fill_args(mode, c, spec, &a, blk_n, blk_k);
  a.planar = ICS_FFT_PL_ALL;
  static std::atomic<bool> configured[2][ICS_MAX_DEVICES];
  const int dev = ics_current_device();
  int grid = ics_device_cus(dev);
  if (const int mw = ics_debug().max_wgs.load(std::memory_order_relaxed); mw > 0 && grid > mw) grid = mw;
  if (grid > a.nunits) grid = a.nunits;
  auto k0 = icsfft::k_conv_fft_blk<0>;
  auto k1 = icsfft::k_conv_fft_blk<1>;
  auto kern = mode == 0 ? k0 : k1;
  if (hipError_t e = ics_configure_lds(configured[mode], dev, kern, ICS_FFT_LDS_BYTES); e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(ICS_FFT_THREADS), ICS_FFT_LDS_BYTES, s, a);
  return hipGetLastError();
}
// A12 + A13 with tap blocks: one launch of k_gradk_fft per block of lags (the residual's transform is repeated per block: the running sums of
// several blocks do not fit the registers); partial: ics_gradk_fft_blocks() * blk_k^2 floats
hipError_t ics_launch_gradk_fft_blk(const float* u, const float* e, const IcsGeom& g, int blk_n, int blk_k, float* partial, float* gradk, hipStream_t s) {
  IcsConvArgs c;
  memset(&c, 0, sizeof c);
  c.g = g; c.in = u; c.f = e; c.out = const_cast<float*>(e); c.u = u; c.ut = u;
  static std::atomic<bool> configured[ICS_MAX_DEVICES];
  const int dev = ics_current_device();
  auto kern = icsfft::k_gradk_fft<0>;
  if (hipError_t err = ics_configure_lds(configured, dev, kern, ICS_FFT_LDS_BYTES); err != hipSuccess) return err;
  for (int qy = 0; qy < blk_n; ++qy)
    for (int qx = 0; qx < blk_n; ++qx) {
      IcsFftArgs a;
      ics_conv_fft_fill_args(0, c, nullptr, &a, blk_n, blk_k);
      a.planar = ICS_FFT_PL_ALL;
      a.lag_y = qy * blk_k; a.lag_x = qx * blk_k;
      int grid = ics_gradk_fft_blocks(ics_device_cus(dev));
      const int npairs = (a.ntiles + 1) / 2;
      if (grid > 3 * npairs) grid = 3 * npairs;
      hipLaunchKernelGGL(kern, dim3(grid), dim3(ICS_FFT_THREADS), ICS_FFT_LDS_BYTES, s, a, partial);
      hipLaunchKernelGGL(icsfft::k_gradk_fft_reduce_blk, dim3((3 * blk_k * blk_k + 3) / 4), dim3(256), 0, s, partial, grid, g.K, blk_k, a.lag_y, a.lag_x, gradk);
    }
  return hipGetLastError();
}
#endif
