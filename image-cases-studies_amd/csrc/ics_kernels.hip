// ics_kernels.hip -- image update, PSF gradient (MFMA) and PSF step of the RL/MM loop, gfx950.
#include <stdlib.h>

#include "ics_kernels.h"
#include "ics_tv.h"

#ifndef ICS_UPDATE_U
#define ICS_UPDATE_U 4   /* wave segments (1 KiB per frame each) per loop iteration of k_update_rows */
#endif
#ifndef ICS_PSF_THREADS
#define ICS_PSF_THREADS 1024
#endif
#ifndef ICS_UPDATE_U_TV
#define ICS_UPDATE_U_TV 4   /* the same for the active MM-TV form (five frames read, two written) */
#endif
#ifndef ICS_UPDATE_NT
#define ICS_UPDATE_NT 15  /* streaming (nt) loads in k_update_rows: bit 0 u, 1 ut, 2 g, 3 f */
#endif
#ifndef ICS_GRADK_WAVES
#define ICS_GRADK_WAVES 4   /* waves per k_gradk workgroup (4 or 8: measured equal, 0.28 ms at 4096^2) */
#endif

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t key_of(float f) {
  if (f != f) return 0xFFC00000u;  // canonical +NaN: propagates through an integer max like np.amax
  return ics_f2key(f);
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) { uint32_t o = (uint32_t)__shfl_xor((int)v, off, 64); v = v > o ? v : o; }
  return v;
}
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) { uint32_t o = (uint32_t)__shfl_xor((int)v, off, 64); v = v < o ? v : o; }
  return v;
}

// =================================================================================================
// A5 + A6 + A8 + A10 (lib/deconvolution.pyx:499-552), one pass over the u-frame.
//   DoF   = ((gradu - image)/(gradu + image))^2 [ / lambd when non-blind ]      (:499-502, interior)
//   g     = lambd*gradu + (u - ut)/2.                                           (:519, else-branch)
//   dt_k  = step*(max u_k + 0)/(max|g_k| + 1e-15)                               (:524, 1/(M*N) == 0)
//   u    -= dt_k * g                                                            (:531)
//   u     = (1 - DoF)*u + DoF*image   on the interior                           (:552)
// A9 (:534-549) subtracts exactly zero from `image` and is therefore omitted.
// Every operation is rounded separately (__f*_rn) like the reference's C / numpy float32 code.
// Memory-bound: 4 frame reads + 1 write (60 B/px), dwordx4 on the flattened x*3+c axis.
// =================================================================================================
__global__ __launch_bounds__(256) void k_update(IcsUpdateArgs a) {
  const IcsGeom& G = a.geo;
  const int ngx = G.tiles_x * 16;  // groups of 4 px per row
  const long total = (long)G.uM * ngx;
  float dt[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float maxu = ics_key2f(a.red[ICS_RED_MAXU + c]);
    const float maxg = ics_key2f(a.red[ICS_RED_MAXG + c]);
    dt[c] = __fdiv_rn(__fmul_rn(a.step, maxu), __fadd_rn(maxg, 1e-15f));
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      a.scal[ICS_SC_DT + c] = dt[c]; a.scal[ICS_SC_MAXU + c] = maxu; a.scal[ICS_SC_MAXG + c] = maxg;
    }
  }
  float dt2[3] = {0.f, 0.f, 0.f};
  if (a.tv_kind == 1) {  // pyx:548: dt = step*(max image_k + 0)/(max|gradu_k| + 1e-15) with gradu = T
#pragma unroll
    for (int c = 0; c < 3; ++c)
      dt2[c] = __fdiv_rn(__fmul_rn(a.step, ics_key2f(a.red[ICS_RED_MAXF + c])), __fadd_rn(ics_key2f(a.red[ICS_RED_MAXT + c]), 1e-15f));
  }
  uint32_t kmin = 0xFFFFFFFFu, kmax = 0u, knan = 0u;
  const float lambd = a.lambd;
  for (long gid = (long)blockIdx.x * 256 + threadIdx.x; gid < total; gid += (long)gridDim.x * 256) {
    const int y = (int)(gid / ngx);
    const int xp = 4 * (int)(gid - (long)y * ngx);
    if (xp >= G.uN) continue;
    const ptrdiff_t o = (ptrdiff_t)y * G.pitch + 3 * xp;
    float uv[12], tv[12], gv[12], fv[12], Tv[12];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const f32x4 p = reinterpret_cast<const f32x4*>(a.u + o)[j];
      const f32x4 q = reinterpret_cast<const f32x4*>(a.ut + o)[j];
      const f32x4 r = reinterpret_cast<const f32x4*>(a.g + o)[j];
      const f32x4 s = reinterpret_cast<const f32x4*>(a.f + o)[j];
      uv[4*j] = p.x; uv[4*j+1] = p.y; uv[4*j+2] = p.z; uv[4*j+3] = p.w;
      tv[4*j] = q.x; tv[4*j+1] = q.y; tv[4*j+2] = q.z; tv[4*j+3] = q.w;
      if (a.tv) { const f32x4 w = reinterpret_cast<const f32x4*>(a.tv + o)[j]; Tv[4*j] = w.x; Tv[4*j+1] = w.y; Tv[4*j+2] = w.z; Tv[4*j+3] = w.w; }
      gv[4*j] = r.x; gv[4*j+1] = r.y; gv[4*j+2] = r.z; gv[4*j+3] = r.w;
      fv[4*j] = s.x; fv[4*j+1] = s.y; fv[4*j+2] = s.z; fv[4*j+3] = s.w;
    }
    const bool yin = (y >= G.pad) && (y < G.pad + G.M);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int x = xp + p;
      const bool inside = yin && (x >= G.pad) && (x < G.pad + G.N);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int i = 3 * p + c;
        float g;
        if (a.tv_kind >= 2)                                                    // PAM: the back-projection pass wrote G = T + lambd*gradu
          g = gv[i];
        else if (a.tv_kind == 1 && y >= 1 && y <= G.uM - 2 && x >= 1 && x <= G.uN - 2)   // pyx:517 (TV_ut_L1 != 0 and TV_u_L1 != 0)
          g = (float)(((double)Tv[i] + (double)__fmul_rn(lambd, gv[i])) + (double)__fsub_rn(uv[i], tv[i]) / 4.0);
        else
          g = __fadd_rn(__fmul_rn(lambd, gv[i]), __fmul_rn(__fsub_rn(uv[i], tv[i]), 0.5f));
        float un = __fsub_rn(uv[i], __fmul_rn(dt[c], g));
        if (inside && a.tv_kind < 2) {   // (PAM has no DoF blend)
          const float d = ics_dof_ratio(gv[i], fv[i]);
          float D = __fmul_rn(d, d);
          if (!a.blind) D = __fdiv_rn(D, lambd);
          if (a.tv_kind == 1) {  // pyx:549: image -= dt*gradu/lambd, then the blend uses the updated image
            fv[i] = __fsub_rn(fv[i], __fdiv_rn(__fmul_rn(dt2[c], Tv[i]), lambd));
            a.f_rw[o + i] = fv[i];
          }
          un = __fadd_rn(__fmul_rn(__fsub_rn(1.0f, D), un), __fmul_rn(D, fv[i]));
          if (a.want_dof) {
            if (D != D) knan = 1u;
            else { const uint32_t k = ics_f2key(D); kmin = kmin < k ? kmin : k; kmax = kmax > k ? kmax : k; }
          }
        }
        uv[i] = un;
      }
    }
    if (xp + 3 < G.uN) {
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const f32x4 w = {uv[4*j], uv[4*j+1], uv[4*j+2], uv[4*j+3]};
        reinterpret_cast<f32x4*>(a.u_out + o)[j] = w;
      }
    } else {
#pragma unroll
      for (int p = 0; p < 4; ++p)
        if (xp + p < G.uN) { a.u_out[o + 3*p] = uv[3*p]; a.u_out[o + 3*p + 1] = uv[3*p+1]; a.u_out[o + 3*p + 2] = uv[3*p+2]; }
    }
  }
  if (a.want_dof) {  // wave shuffle -> one atomic per wave (grid is capped, so a few thousand atomics)
    kmin = wave_min_u32(kmin); kmax = wave_max_u32(kmax); knan = wave_max_u32(knan);
    if ((threadIdx.x & 63) == 0) {
      atomicMin(a.dofkeys + 0, kmin); atomicMax(a.dofkeys + 1, kmax);
      if (knan) atomicOr(a.dofkeys + 2, 1u);
    }
  }
}

// PAM total-variation gradient (oracle/rl_ext_oracle.py pam_tv_term): T = -div(p) at one float.
// n = the 3 x 6-px neighbourhood rows (y-1, y, y+1) of u, i = index of the centre float, c = its channel.
//   forward differences d_x = u[y+1]-u[y], d_y = u[x+1]-u[x]; backward-difference divergence
//   isotropic:      p = d / sqrt(dx^2 + dy^2 + eps^2)            (per channel)
//   collaborative:  p_d,c = [c == first argmax_c' |d_d u_c'|] * d / sqrt(d^2 + eps^2)   (L-inf over channels)
__device__ __forceinline__ float pam_term(const float (&n)[3][20], int i, int c, float eps, bool collaborative) {
  const float e2 = __fmul_rn(eps, eps);
  if (!collaborative) {
    const float dx0 = __fsub_rn(n[2][i], n[1][i]), dy0 = __fsub_rn(n[1][i + 3], n[1][i]);          // at (y, x)
    const float dxu = __fsub_rn(n[1][i], n[0][i]), dyu = __fsub_rn(n[0][i + 3], n[0][i]);          // at (y-1, x)
    const float dxl = __fsub_rn(n[2][i - 3], n[1][i - 3]), dyl = __fsub_rn(n[1][i], n[1][i - 3]);  // at (y, x-1)
    const float n0 = __fsqrt_rn(__fadd_rn(__fadd_rn(__fmul_rn(dx0, dx0), __fmul_rn(dy0, dy0)), e2));
    const float nu_ = __fsqrt_rn(__fadd_rn(__fadd_rn(__fmul_rn(dxu, dxu), __fmul_rn(dyu, dyu)), e2));
    const float nl = __fsqrt_rn(__fadd_rn(__fadd_rn(__fmul_rn(dxl, dxl), __fmul_rn(dyl, dyl)), e2));
    const float div = __fadd_rn(__fsub_rn(__fdiv_rn(dx0, n0), __fdiv_rn(dxu, nu_)), __fsub_rn(__fdiv_rn(dy0, n0), __fdiv_rn(dyl, nl)));
    return -div;
  }
  const int b = i - c;  // channel 0 of this pixel
  float px0 = 0.f, pxu = 0.f, py0 = 0.f, pyl = 0.f;
  {  // x-direction at (y, x) and (y-1, x)
    float d[3], du[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { d[k] = __fsub_rn(n[2][b + k], n[1][b + k]); du[k] = __fsub_rn(n[1][b + k], n[0][b + k]); }
    int s = 0, su = 0;
#pragma unroll
    for (int k = 1; k < 3; ++k) { if (__builtin_fabsf(d[k]) > __builtin_fabsf(d[s])) s = k; if (__builtin_fabsf(du[k]) > __builtin_fabsf(du[su])) su = k; }
    if (s == c) px0 = __fdiv_rn(d[c], __fsqrt_rn(__fadd_rn(__fmul_rn(d[c], d[c]), e2)));
    if (su == c) pxu = __fdiv_rn(du[c], __fsqrt_rn(__fadd_rn(__fmul_rn(du[c], du[c]), e2)));
  }
  {  // y-direction at (y, x) and (y, x-1)
    float d[3], dl[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { d[k] = __fsub_rn(n[1][b + k + 3], n[1][b + k]); dl[k] = __fsub_rn(n[1][b + k], n[1][b + k - 3]); }
    int s = 0, sl = 0;
#pragma unroll
    for (int k = 1; k < 3; ++k) { if (__builtin_fabsf(d[k]) > __builtin_fabsf(d[s])) s = k; if (__builtin_fabsf(dl[k]) > __builtin_fabsf(dl[sl])) sl = k; }
    if (s == c) py0 = __fdiv_rn(d[c], __fsqrt_rn(__fadd_rn(__fmul_rn(d[c], d[c]), e2)));
    if (sl == c) pyl = __fdiv_rn(dl[c], __fsqrt_rn(__fadd_rn(__fmul_rn(dl[c], dl[c]), e2)));
  }
  return -__fadd_rn(__fsub_rn(px0, pxu), __fsub_rn(py0, pyl));
}

// =================================================================================================
// Active MM-TV (build-defined extension; oracle/rl_ext_oracle.py): the regulariser term of pyx:517/:543,
//   T = float( div/TV_u_L1/TV_ut_L1/2. + div/TV_u_L2/TV_ut_L2/2. )  on the interior of the u-frame, 0 on its border,
// with TV(.) the order-2 stencil of pyx:137-189 (norm 1 and 2), `div` in its norm-2 scaling (the second TV call of
// pyx:495-496 overwrites it) and TV_ut evaluated from the majoriser ut.  Also reduces max|T_k| and max image_k
// for the image step of pyx:548.  One lane = 4 pixels; the 3x6-px neighbourhoods of u and ut come as 3 rows of
// 5 unaligned dwordx4 loads each (served by L1/L2).
// =================================================================================================
// One thread = 4 pixels x TVSEG rows, walking down with a three-row window in registers (each row of u / ut is loaded
// once per thread instead of three times; KIND is a template parameter so that the PAM kinds do not load ut at all).
// (KIND 1 evaluates four stencils with IEEE divisions and square roots per value: bound by VALU latency, it ran slower with
//  the 164 registers of the walking window -- 0.50 vs 0.44 ms -- and keeps one row per thread.)
// The MM-TV term of one value, fast form (round 3).  tv_term of oracle/rl_ext_oracle.py evaluates the order-2 stencil four times
// (u and ut, norm 1 and norm 2: ics_tv_point) with IEEE divisions and square roots -- ~300 vector instructions per value, which
// made k_tvterm<1> VALU-bound at 2.5x the time of the memory traffic it causes.  The four evaluations share their second
// differences; those stay exact (double sums of floats, rounded once, as the reference computes them: the term is largest where
// the image is flattest, i.e. where the differences cancel), everything behind them uses v_rcp_f32 / v_sqrt_f32 (1 ulp).  Within
// ~1e-6 of the IEEE form relative to max |T| (gate 1e-5, tests/test_tv_mode.py); ICS_PAM_EXACT=1 / debug switch pam_exact (one switch for all
// three extended kinds, tv_mode 1, 2 and 3: ics_launch_tvterm) selects the IEEE form.  n = rows (y-1, y, y+1) of the 20-float windows, i = index of the centre float.
#ifndef ICS_TVMM_SEG
#define ICS_TVMM_SEG 8
#endif
#ifndef ICS_TVMM_F64_DIFF
#define ICS_TVMM_F64_DIFF 0   /* second differences as double sums (round 3's first form) */
#endif
struct IcsStencil2 { float udx, udy, udd, uda; };
__device__ __forceinline__ IcsStencil2 ics_second_differences(const float (&n)[3][20], int i) {
#if ICS_TVMM_F64_DIFF
  const double m2c = -2.0 * (double)n[1][i];
  const float inv = 0.707106769f;   // 1 / 1.41421354f rounded to float
  IcsStencil2 s;
  s.udx = (float)((m2c + (double)n[0][i]) + (double)n[2][i]);
  s.udy = (float)((m2c + (double)n[1][i - 3]) + (double)n[1][i + 3]);
  s.udd = (float)((m2c + (double)n[0][i - 3]) + (double)n[2][i + 3]) * inv;
  s.uda = (float)((m2c + (double)n[0][i + 3]) + (double)n[2][i - 3]) * inv;
  return s;
#else
  // (a - c) + (b - c) in fp32: where the neighbours lie within a factor two of the centre -- every place where the differences
  // cancel, i.e. where the term is large -- both differences are exact (Sterbenz) and the sum is rounded once, which is what the
  // double form above returns; elsewhere the result is within an ulp of it.  A third fewer vector instructions per value.
  const float c = n[1][i];
  const float inv = 0.707106769f;   // 1 / 1.41421354f rounded to float
  IcsStencil2 s;
  s.udx = __fadd_rn(__fsub_rn(n[0][i], c), __fsub_rn(n[2][i], c));
  s.udy = __fadd_rn(__fsub_rn(n[1][i - 3], c), __fsub_rn(n[1][i + 3], c));
  s.udd = __fadd_rn(__fsub_rn(n[0][i - 3], c), __fsub_rn(n[2][i + 3], c)) * inv;
  s.uda = __fadd_rn(__fsub_rn(n[0][i + 3], c), __fsub_rn(n[2][i - 3], c)) * inv;
  return s;
#endif
}
__device__ __forceinline__ float ics_tv_mm_term_fast(const float (&nu)[3][20], const float (&nt)[3][20], int i, float eps) {
  const float a1 = 6.82842731f, a2 = 4.82842731f;            // 4 (1 + 1/sqrt 2), 2 (1 + sqrt 2)
  const IcsStencil2 su = ics_second_differences(nu, i), st = ics_second_differences(nt, i);
  const float e2 = eps * eps;
  const float u1 = ((__builtin_fabsf(su.udx) + __builtin_fabsf(su.udy)) + eps) + ((__builtin_fabsf(su.udd) + __builtin_fabsf(su.uda)) + eps);   // x adjust1
  const float u2 = __builtin_amdgcn_sqrtf(su.udx * su.udx + su.udy * su.udy + e2) + __builtin_amdgcn_sqrtf(su.udd * su.udd + su.uda * su.uda + e2);   // x adjust2
  const float t1 = ((__builtin_fabsf(st.udx) + __builtin_fabsf(st.udy)) + eps) + ((__builtin_fabsf(st.udd) + __builtin_fabsf(st.uda)) + eps);
  const float t2 = __builtin_amdgcn_sqrtf(st.udx * st.udx + st.udy * st.udy + e2) + __builtin_amdgcn_sqrtf(st.udd * st.udd + st.uda * st.uda + e2);
  const float d = ((-su.udx - su.udy) - su.udd) - su.uda;      // x adjust2 (div carries the norm-2 scaling, pyx:495-496)
  // T = (div / TV_u_L1) / TV_ut_L1 / 2 + (div / TV_u_L2) / TV_ut_L2 / 2 with div = d / a2, TV_x_L1 = x1 / a1, TV_x_L2 = x2 / a2
  const float q1 = d * (a1 * a1 / a2) * __builtin_amdgcn_rcpf(u1 * t1);
  const float q2 = d * a2 * __builtin_amdgcn_rcpf(u2 * t2);
  return 0.5f * (q1 + q2);
}

template <int KIND, bool FAST = false>
__global__ __launch_bounds__(256) void k_tvterm(IcsTvTermArgs a, int seg_fast) {
  // rows a thread walks (the fast MM form walks like the PAM kinds, 2 ... ICS_TVMM_SEG rows chosen by frame size: ics_launch_tvterm)
  const int TVSEG = KIND == 1 ? (FAST ? seg_fast : 1) : 16;
  const IcsGeom& G = a.geo;
  const int ngx = G.tiles_x * 16;
  const int nseg = (G.uM + TVSEG - 1) / TVSEG;
  const long total = (long)nseg * ngx;
  typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
  float mt[3] = {0.f, 0.f, 0.f}, mf[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
  bool nan_t[3] = {false, false, false}, nan_f[3] = {false, false, false}, any_f = false;
  const float eps = a.epsilon;
  auto load_row = [&](const float* base, int y, int xp, float (&row)[20]) {
    const ptrdiff_t o = (ptrdiff_t)y * G.pitch + 3 * (xp - 1);
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const f32x4u p = *reinterpret_cast<const f32x4u*>(base + o + 4 * j);
      row[4*j] = p.x; row[4*j+1] = p.y; row[4*j+2] = p.z; row[4*j+3] = p.w;
    }
  };
  for (long gid = (long)blockIdx.x * 256 + threadIdx.x; gid < total; gid += (long)gridDim.x * 256) {
    const int seg = (int)(gid / ngx);
    const int xp = 4 * (int)(gid - (long)seg * ngx);
    if (xp >= G.uN) continue;
    const int ybeg = seg * TVSEG, yend = ybeg + TVSEG < G.uM ? ybeg + TVSEG : G.uM;
    float nu[3][20], nt[3][20];
    load_row(a.u, ybeg - 1, xp, nu[0]); load_row(a.u, ybeg, xp, nu[1]);
    if (KIND == 1) { load_row(a.ut, ybeg - 1, xp, nt[0]); load_row(a.ut, ybeg, xp, nt[1]); }
    for (int y = ybeg; y < yend; ++y) {
    load_row(a.u, y + 1, xp, nu[2]);
    if (KIND == 1) load_row(a.ut, y + 1, xp, nt[2]);
    const ptrdiff_t o = (ptrdiff_t)y * G.pitch + 3 * xp;
    float T[12];
    const bool yact = (y >= 1) && (y <= G.uM - 2);
    const bool yin = (y >= G.pad) && (y < G.pad + G.M);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int x = xp + p;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int i = 3 * (p + 1) + c;
        float t = 0.f;
        if (KIND >= 2) {
          if (yact && x >= 1 && x <= G.uN - 2) t = pam_term(nu, i, c, eps, KIND == 3);
        } else if (FAST) {   // branch-free: the windows of border values lie in the frame's apron, their term is discarded
          const float tt = ics_tv_mm_term_fast(nu, nt, i, eps);
          t = (yact && x >= 1 && x <= G.uN - 2) ? tt : 0.f;
        } else if (yact && x >= 1 && x <= G.uN - 2) {
          const IcsTvOut u1 = ics_tv_point(nu[1][i], nu[0][i], nu[2][i], nu[1][i-3], nu[1][i+3], nu[0][i-3], nu[2][i+3], nu[0][i+3], nu[2][i-3], eps, 2, 1);
          const IcsTvOut u2 = ics_tv_point(nu[1][i], nu[0][i], nu[2][i], nu[1][i-3], nu[1][i+3], nu[0][i-3], nu[2][i+3], nu[0][i+3], nu[2][i-3], eps, 2, 2);
          const IcsTvOut t1 = ics_tv_point(nt[1][i], nt[0][i], nt[2][i], nt[1][i-3], nt[1][i+3], nt[0][i-3], nt[2][i+3], nt[0][i+3], nt[2][i-3], eps, 2, 1);
          const IcsTvOut t2 = ics_tv_point(nt[1][i], nt[0][i], nt[2][i], nt[1][i-3], nt[1][i+3], nt[0][i-3], nt[2][i+3], nt[0][i+3], nt[2][i-3], eps, 2, 2);
          const double d1 = (double)__fdiv_rn(__fdiv_rn(u2.div, u1.out), t1.out) / 2.0;
          const double d2 = (double)__fdiv_rn(__fdiv_rn(u2.div, u2.out), t2.out) / 2.0;
          t = (float)(d1 + d2);
        }
        T[3*p+c] = t;
        if (x < G.uN) { mt[c] = __builtin_fmaxf(mt[c], __builtin_fabsf(t)); nan_t[c] |= (t != t); }
      }
    }
    if (yin) {  // max image_k over the M x N image
      float fv[12];
#pragma unroll
      for (int j = 0; j < 3; ++j) { const f32x4 q = reinterpret_cast<const f32x4*>(a.f + o)[j]; fv[4*j] = q.x; fv[4*j+1] = q.y; fv[4*j+2] = q.z; fv[4*j+3] = q.w; }
#pragma unroll
      for (int p = 0; p < 4; ++p)
        if (xp + p >= G.pad && xp + p < G.pad + G.N) {
#pragma unroll
          for (int c = 0; c < 3; ++c) { mf[c] = __builtin_fmaxf(mf[c], fv[3*p+c]); nan_f[c] |= (fv[3*p+c] != fv[3*p+c]); }
          any_f = true;
        }
    }
    if (xp + 3 < G.uN) {
#pragma unroll
      for (int j = 0; j < 3; ++j) { const f32x4 w = {T[4*j], T[4*j+1], T[4*j+2], T[4*j+3]}; reinterpret_cast<f32x4*>(a.tv + o)[j] = w; }
    } else {
#pragma unroll
      for (int p = 0; p < 4; ++p)
        if (xp + p < G.uN) { a.tv[o + 3*p] = T[3*p]; a.tv[o + 3*p + 1] = T[3*p+1]; a.tv[o + 3*p + 2] = T[3*p+2]; }
    }
#pragma unroll
    for (int k = 0; k < 20; ++k) {
      nu[0][k] = nu[1][k]; nu[1][k] = nu[2][k];
      if (KIND == 1) { nt[0][k] = nt[1][k]; nt[1][k] = nt[2][k]; }
    }
    }
  }
  // maxima: wave -> workgroup (LDS) -> one atomic per workgroup and value (per-wave atomics on six words were most of this kernel's
  // time on small frames)
  __shared__ uint32_t shk[4][6];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    uint32_t kt = nan_t[c] ? 0xFFC00000u : ics_f2key(mt[c]);
    uint32_t kf = nan_f[c] ? 0xFFC00000u : (any_f ? ics_f2key(mf[c]) : 0u);
    kt = wave_max_u32(kt); kf = wave_max_u32(kf);
    if ((threadIdx.x & 63) == 0) { shk[threadIdx.x >> 6][c] = kt; shk[threadIdx.x >> 6][3 + c] = kf; }
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    const int v = threadIdx.x;
    uint32_t k = shk[0][v];
#pragma unroll
    for (int w = 1; w < 4; ++w) k = k > shk[w][v] ? k : shk[w][v];
    uint32_t* dst = a.red + (v < 3 ? ICS_RED_MAXT + v : ICS_RED_MAXF + (v - 3));
    if (k > *dst) atomicMax(dst, k);
  }
}

// =================================================================================================
// PAM kinds (2 isotropic, 3 collaborative) of the TV term, round 3: the same quantity as pam_term() above -- T = -div(p), p the
// normalised forward differences -- evaluated ONCE per pixel and direction instead of once per use.  pam_term() recomputes the
// three norms (own pixel, pixel above, pixel to the left) of every value with IEEE square roots and divisions: ~100 vector
// instructions per value, which made k_tvterm<2|3> VALU-bound at 0.17 ms for a 4096^2 frame (24 B/px of traffic = 0.07 ms).  Here a
// thread walks down 16 rows with p_x of the row above in registers and takes p_y of the pixel to the left from its own 5-pixel
// window; 1 / sqrt comes from v_rsq_f32 (1 ulp).  ~10 instructions per value; results within ~2e-7 of pam_term() relative to
// max |T| (tests/test_tv_mode.py gates 2e-6 against oracle/rl_ext_oracle.py; these modes have no reference implementation).
// =================================================================================================
template <bool COLLAB, bool PL = false>   // PL: u and T are channel-planar mirrors -- the same arithmetic on the same values, bit for bit
__global__ __launch_bounds__(256) void k_tvterm_pam(IcsTvTermArgs a, int TVSEG /* rows a thread walks: 16 on large frames, fewer where that leaves the chip idle */) {
  const IcsGeom& G = a.geo;
  const int ppitch = PL ? ics_ppitch(G) : 0;
  const size_t plane = PL ? ics_plane_floats(G) : 0;
  const int ngx = G.tiles_x * 16;
  const int nseg = (G.uM + TVSEG - 1) / TVSEG;
  const long total = (long)nseg * ngx;
  typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
  float mt[3] = {0.f, 0.f, 0.f};
  bool nan_t[3] = {false, false, false};
  const float e2 = __fmul_rn(a.epsilon, a.epsilon);
  auto load_row = [&](int y, int xp, float (&row)[20]) {   // pixels xp - 1 .. xp + 5 (20 floats)
    if (PL) {   // per plane: the pixel to the left, the aligned quad, the pixel to the right (row[18], row[19] are never read)
      const ptrdiff_t o = (ptrdiff_t)y * ppitch + xp;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float* pc = a.u + c * plane + o;
        const f32x4 q = *reinterpret_cast<const f32x4*>(pc);
        row[c] = pc[-1]; row[3 + c] = q.x; row[6 + c] = q.y; row[9 + c] = q.z; row[12 + c] = q.w; row[15 + c] = pc[4];
      }
      row[18] = row[19] = 0.f;
      return;
    }
    const ptrdiff_t o = (ptrdiff_t)y * G.pitch + 3 * (xp - 1);
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const f32x4u p = *reinterpret_cast<const f32x4u*>(a.u + o + 4 * j);
      row[4*j] = p.x; row[4*j+1] = p.y; row[4*j+2] = p.z; row[4*j+3] = p.w;
    }
  };
  // p of the row `cur` (with `nxt` below it) at the window pixels j = 0 .. 4 (frame pixels xp - 1 .. xp + 3): px[3 j + c], py[3 j + c]
  auto normalised = [&](const float (&cur)[20], const float (&nxt)[20], float (&px)[15], float (&py)[15], bool want_y) {
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      float dx[3], dy[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) { dx[c] = __fsub_rn(nxt[3*j+c], cur[3*j+c]); dy[c] = __fsub_rn(cur[3*j+c+3], cur[3*j+c]); }
      if (!COLLAB) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float r = __builtin_amdgcn_rsqf(__fadd_rn(__fadd_rn(__fmul_rn(dx[c], dx[c]), __fmul_rn(dy[c], dy[c])), e2));
          px[3*j+c] = __fmul_rn(dx[c], r);
          py[3*j+c] = __fmul_rn(dy[c], r);
        }
      } else {   // only the first channel of largest |difference| carries the term, per direction
        int sx = 0, sy = 0;
#pragma unroll
        for (int k = 1; k < 3; ++k) { if (__builtin_fabsf(dx[k]) > __builtin_fabsf(dx[sx])) sx = k; if (__builtin_fabsf(dy[k]) > __builtin_fabsf(dy[sy])) sy = k; }
        const float vx = sx == 0 ? dx[0] : (sx == 1 ? dx[1] : dx[2]), vy = sy == 0 ? dy[0] : (sy == 1 ? dy[1] : dy[2]);
        const float qx = __fmul_rn(vx, __builtin_amdgcn_rsqf(__fadd_rn(__fmul_rn(vx, vx), e2)));
        const float qy = want_y ? __fmul_rn(vy, __builtin_amdgcn_rsqf(__fadd_rn(__fmul_rn(vy, vy), e2))) : 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) { px[3*j+c] = c == sx ? qx : 0.f; py[3*j+c] = c == sy ? qy : 0.f; }
      }
    }
  };
  for (long gid = (long)blockIdx.x * 256 + threadIdx.x; gid < total; gid += (long)gridDim.x * 256) {
    const int seg = (int)(gid / ngx);
    const int xp = 4 * (int)(gid - (long)seg * ngx);
    if (xp >= G.uN) continue;
    const int ybeg = seg * TVSEG, yend = ybeg + TVSEG < G.uM ? ybeg + TVSEG : G.uM;
    float ra[20], rb[20], rc[20];
    float pxu[15], px0[15], py0[15], pyd[15];
    load_row(ybeg - 1, xp, ra); load_row(ybeg, xp, rb);
    normalised(ra, rb, pxu, pyd, false);                 // p_x of the row above the first one
    for (int y = ybeg; y < yend; ++y) {
      load_row(y + 1, xp, rc);
      normalised(rb, rc, px0, py0, true);
      const ptrdiff_t o = (ptrdiff_t)y * G.pitch + 3 * xp;
      const bool yact = (y >= 1) && (y <= G.uM - 2);
      float T[12];
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const int x = xp + p;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const int i = 3 * (p + 1) + c;                   // window index of (x, c); (x - 1, c) is i - 3
          float t = 0.f;
          if (yact && x >= 1 && x <= G.uN - 2) t = -__fadd_rn(__fsub_rn(px0[i], pxu[i]), __fsub_rn(py0[i], py0[i - 3]));
          T[3*p+c] = t;
          if (x < G.uN) { mt[c] = __builtin_fmaxf(mt[c], __builtin_fabsf(t)); nan_t[c] |= (t != t); }
        }
      }
      if (PL) {   // (the quad's pixels beyond uN lie in the mirror's apron: T = 0 there by the x <= uN - 2 test above)
        const ptrdiff_t op = (ptrdiff_t)y * ppitch + xp;
#pragma unroll
        for (int c = 0; c < 3; ++c) { const f32x4 w = {T[c], T[3 + c], T[6 + c], T[9 + c]}; *reinterpret_cast<f32x4*>(a.tv + c * plane + op) = w; }
      } else if (xp + 3 < G.uN) {
#pragma unroll
        for (int j = 0; j < 3; ++j) { const f32x4 w = {T[4*j], T[4*j+1], T[4*j+2], T[4*j+3]}; reinterpret_cast<f32x4*>(a.tv + o)[j] = w; }
      } else {
#pragma unroll
        for (int p = 0; p < 4; ++p)
          if (xp + p < G.uN) { a.tv[o + 3*p] = T[3*p]; a.tv[o + 3*p + 1] = T[3*p+1]; a.tv[o + 3*p + 2] = T[3*p+2]; }
      }
#pragma unroll
      for (int k = 0; k < 20; ++k) rb[k] = rc[k];
#pragma unroll
      for (int k = 0; k < 15; ++k) pxu[k] = px0[k];
    }
  }
  // maxima: wave -> workgroup (LDS) -> one atomic per workgroup and channel, skipped when the running maximum is already there
  __shared__ uint32_t shk[4][3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    uint32_t kt = nan_t[c] ? 0xFFC00000u : ics_f2key(mt[c]);
    kt = wave_max_u32(kt);
    if ((threadIdx.x & 63) == 0) shk[threadIdx.x >> 6][c] = kt;
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    const int c = threadIdx.x;
    uint32_t kt = shk[0][c];
#pragma unroll
    for (int w = 1; w < 4; ++w) kt = kt > shk[w][c] ? kt : shk[w][c];
    if (kt > a.red[ICS_RED_MAXT + c]) atomicMax(a.red + ICS_RED_MAXT + c, kt);
  }
}

// =================================================================================================
// A13 (lib/deconvolution.pyx:567-571): gradk = convolve(rot180(u), error, "valid"), i.e.
//     gradk[a, b, c] = sum_{y,x} E[y, x, c] * U[y + pad - a, x + pad - b, c]      (u-frame coords)
// 3*K*K outputs, each an M*N-term dot product.  Unlike the image convolutions this IS a dense
// contraction: for one image row y and channel c,
//     D[a][b] += sum_k A[a][k] * B[k][b],   A[a][k] = U[y+pad-a][xk],  B[k][b] = E[y][xk - pad + b]
// with the long pixel axis as K, so it runs on the fp32 matrix cores (v_mfma_f32_16x16x4_f32, exact
// fp32 == an fmaf chain) at (K/16)^2 = 88 % useful work for a 15x15 PSF.  rot180(u) (A12, pyx:567)
// is never materialised: the flip is the minus sign in the indices.
// Workgroups are persistent (grid-stride over 64x32-px tiles, next tile prefetched into registers
// during the MFMA phase) and keep their 16x16 accumulators in registers across tiles; one partial block per workgroup is written at the end and reduced in
// double, in a fixed order, by k_gradk_reduce (deterministic, no float atomics).
// =================================================================================================

template <int NB, int NW = 4>
struct GradkCfg {
  static constexpr int TW = 64, TH = 32, NT = 16 * NB;
  static constexpr int NTH = 64 * NW;              // threads per workgroup (NW waves, TH/NW tile rows each)
  static constexpr int UROWS = TH + NT - 1;
  static constexpr int LWU = 3 * TW + 2;          // == 2 (mod 32): the 16 rows x 2 k of an A read hit 32 banks
  static constexpr int EPX = TW + 24 * NB;        // E pixels staged per row: [x0 - 8NB, x0 + 64 + 16NB)
  static constexpr int LWE = 3 * EPX;
  static constexpr size_t LDS_FLOATS = (size_t)UROWS * LWU + (size_t)TH * LWE;
  static constexpr size_t RED_FLOATS = (size_t)NW * NB * NB * 256;   // one channel at a time
  static constexpr size_t LDS_BYTES = 4 * (LDS_FLOATS > RED_FLOATS ? LDS_FLOATS : RED_FLOATS);
};

template <int NB, int NW>
__global__ __launch_bounds__(64 * NW) void k_gradk(IcsGradkArgs a) {
  using C = GradkCfg<NB, NW>;
  constexpr int NTH = C::NTH;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* ul = lds;
  float* el = lds + C::UROWS * C::LWU;
  const IcsGeom& G = a.geo;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = lane & 15, q = lane >> 4;
  const int ntx = G.tiles_x, nty = G.tiles_y * (ICS_TILE / C::TH);
  const int pad = G.pad, pitch = G.pitch;

  f32x4 acc[3][NB][NB];
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j) acc[c][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // Staging is split (issue early / write late): the global loads of tile t+1 are issued right after
  // the LDS image of tile t is complete and stay in flight during its MFMA phase; they are written to
  // LDS only after the barrier that ends the phase.  The kernel needs few registers besides, so the
  // ~70 staging VGPRs are free and HBM/L2 latency disappears behind the matrix pipe.
  constexpr int W4 = 3 * C::TW / 4, NU4 = C::UROWS * W4, NUIT = (NU4 + NTH - 1) / NTH;
  constexpr int E4 = C::LWE / 4, NE4 = C::TH * E4, NEIT = (NE4 + NTH - 1) / NTH;
  f32x4 pu[NUIT], pe[NEIT];  // native vectors (HIP's float4 struct kept the array in scratch)
  // (macros, not lambdas: capturing the register arrays by reference would push them to scratch)
#define GK_PREFETCH(T)                                                                              \
  {                                                                                                 \
    const int px0 = ((T) % ntx) * C::TW, py0 = ((T) / ntx) * C::TH;                                 \
    /* U rows [y0 + pad - NT + 1, y0 + pad + TH), px [x0, x0 + 64) */                               \
    const float* src = a.u + (ptrdiff_t)(py0 + pad - C::NT + 1) * pitch + 3 * px0;                  \
    _Pragma("unroll") for (int k = 0; k < NUIT; ++k) {                                              \
      int v = tid + k * NTH; v = v < NU4 ? v : NU4 - 1;                                             \
      const int row = v / W4, c4 = v - row * W4;                                                    \
      pu[k] = *reinterpret_cast<const f32x4*>(src + (ptrdiff_t)row * pitch + 4 * c4);               \
    }                                                                                               \
    /* E rows [y0, y0 + TH), px [x0 - 8NB, x0 + 64 + 16NB) */                                       \
    const float* srce = a.e + (ptrdiff_t)py0 * pitch + 3 * (px0 - 8 * NB);                          \
    _Pragma("unroll") for (int k = 0; k < NEIT; ++k) {                                              \
      int v = tid + k * NTH; v = v < NE4 ? v : NE4 - 1;                                             \
      const int row = v / E4, c4 = v - row * E4;                                                    \
      pe[k] = *reinterpret_cast<const f32x4*>(srce + (ptrdiff_t)row * pitch + 4 * c4);              \
    }                                                                                               \
  }

  // NB = 4 (PSF sizes 49 .. 63) keeps 3 x 16 accumulator blocks = 192 registers: the staging registers of a prefetched tile do
  // not fit beside them (77 spills at 512 VGPRs), so there the rows of a tile are requested when it starts
  constexpr bool PREFETCH = NB < 4;
  const int ntiles = ntx * nty;
  if (PREFETCH) {
    const int tfirst = (int)blockIdx.x < ntiles ? (int)blockIdx.x : ntiles - 1;
    GK_PREFETCH(tfirst)
  }
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    if (!PREFETCH) GK_PREFETCH(t)
    __syncthreads();  // previous tile fully consumed
#pragma unroll
    for (int k = 0; k < NUIT; ++k) {
      const int v = tid + k * NTH;
      if (v < NU4) {
        const int row = v / W4, c4 = v - row * W4;
        f32x2* d = reinterpret_cast<f32x2*>(ul + row * C::LWU + 4 * c4);  // LWU rows are only 8-B aligned
        d[0] = pu[k].xy; d[1] = pu[k].zw;
      }
    }
#pragma unroll
    for (int k = 0; k < NEIT; ++k) {
      const int v = tid + k * NTH;
      if (v < NE4) {
        const int row = v / E4, c4 = v - row * E4;
        *reinterpret_cast<f32x4*>(el + row * C::LWE + 4 * c4) = pe[k];
      }
    }
    __syncthreads();
    if (PREFETCH) {  // unconditional (clamped) so that the staging registers stay plain SSA values, not scratch
      const int tnext = t + (int)gridDim.x < ntiles ? t + (int)gridDim.x : ntiles - 1;
      GK_PREFETCH(tnext)
    }
#undef GK_PREFETCH
    // wave w owns tile rows [w*TH/NW, (w+1)*TH/NW)
    for (int yy = wave * (C::TH / NW); yy < (wave + 1) * (C::TH / NW); ++yy) {
      const float* arow = ul + (yy + C::NT - 1 - m) * C::LWU + 3 * q;           // + 12*xk + c, - 16*ab rows
      const float* brow = el + yy * C::LWE + 3 * (q + m + 8 * NB - pad);       // + 12*xk + c, + 48*bb
#pragma unroll 4
      for (int xk = 0; xk < C::TW / 4; ++xk) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          float av[NB], bv[NB];
#pragma unroll
          for (int i = 0; i < NB; ++i) av[i] = arow[12 * xk + c - 16 * i * C::LWU];
#pragma unroll
          for (int j = 0; j < NB; ++j) bv[j] = brow[12 * xk + c + 48 * j];
#pragma unroll
          for (int i = 0; i < NB; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j)
              acc[c][i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[j], acc[c][i][j], 0, 0, 0);
        }
      }
    }
  }
  // ---- cross-wave reduction (fixed order) and partial write, one channel per pass (bounds the LDS) --------
  float* red = lds;  // [wave][ab][bb][256]: element (row = 4*q + j, col = m) at [j*64 + lane]
  constexpr int PER = NB * NB * 256;
  float* dst = a.partial + (size_t)blockIdx.x * (3 * C::NT * C::NT);
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          red[((wave * NB + i) * NB + j) * 256 + r * 64 + lane] = acc[c][i][j][r];
    __syncthreads();
    for (int v = tid; v < PER; v += NTH) {
      float s = red[v];
#pragma unroll
      for (int w = 1; w < NW; ++w) s += red[w * PER + v];   // fixed order -> deterministic
      // v = (i*NB + j)*256 + r*64 + l  ->  a = 16 i + 4 (l>>4) + r, b = 16 j + (l & 15)
      const int l = v & 63, r = (v >> 6) & 3, blk = v >> 8;
      const int j = blk % NB, i = blk / NB;
      const int ta = 16 * i + 4 * (l >> 4) + r, tb = 16 * j + (l & 15);
      dst[(c * C::NT + ta) * C::NT + tb] = s;
    }
  }
}

// gradk[a][b][c] = float( sum_blocks double(partial) ), fixed order.  A wave reads 64 consecutive outputs of one workgroup's
// partial block (two cache lines per instruction, eight in flight); the 16 waves of the workgroup take the blocks 16 apart and
// their sums meet in LDS in wave order.  (Round 3: 8.1 -> ~4 us; 32 lanes per output at a 3 KB stride fetched a line per lane.)
// (La x Lb valid taps of the block land at (a0, b0) of the Kf x Kf gradient: the whole gradient is La = Lb = Kf, a0 = b0 = 0; the
//  split gradient of PSF sizes 33 ... 49 reduces four tap blocks, ics_api.hip)
__global__ __launch_bounds__(1024) void k_gradk_reduce(const float* __restrict__ partial, int nblocks, float* __restrict__ gradk, int NT, int La, int Lb,
                                                      int Kf, int a0, int b0) {
  __shared__ double sh[16][64];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int n = 3 * NT * NT;
  const int o = blockIdx.x * 64 + lane;
  const size_t stride = (size_t)n;
  double s = 0.0;
  if (o < n) {
    const float* p = partial + o;
    int b = w;
    for (; b + 7 * 16 < nblocks; b += 8 * 16) {
      float v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = p[(size_t)(b + 16 * k) * stride];
#pragma unroll
      for (int k = 0; k < 8; ++k) s += (double)v[k];
    }
    for (; b < nblocks; b += 16) s += (double)p[(size_t)b * stride];
  }
  sh[w][lane] = s;
  __syncthreads();
  if (w == 0 && o < n) {
    double t = sh[0][lane];
#pragma unroll
    for (int k = 1; k < 16; ++k) t += sh[k][lane];
    const int tb = o % NT, ta = (o / NT) % NT, c = o / (NT * NT);
    if (ta < La && tb < Lb) gradk[((a0 + ta) * Kf + (b0 + tb)) * 3 + c] = (float)t;
  }
}

// =================================================================================================
// A14-A17 (lib/deconvolution.pyx:574-589) + packing of the convolution weights.  One workgroup.
//   dtpsf = step/MK * (max psf + 0) / (max|gradk| + 1e-15)       (:574, global over 3 channels)
//   psf  -= dtpsf * gradk                                         (:577-581)
//   correlation: psf = dstack(mean_c psf x3)                      (:584-585; rebinding: the caller's
//                array keeps the state after the first gradient step, `psf_caller`/`frozen`)
//   clamp < 0 -> 0, divide each channel by its sequential float32 sum   (:587 -> :47-70)
//   psf_rotated = rot180(psf)                                     (:589) -> wconv
// =================================================================================================
// BIG (PSF sizes above 63, ics_big.hip): the working copy of the PSF lives in global memory (a.work; 3 K^2 floats do not fit the LDS
// at K = 127) -- one workgroup, so its barriers order those accesses too -- and no weight tables are packed: k_conv_big reads the
// PSF itself.
template <bool BIG>
__global__ __launch_bounds__(ICS_PSF_THREADS) void k_psf(IcsPsfArgs a) {
  constexpr int NTHR = ICS_PSF_THREADS;   // one workgroup; the passes below are latency chains, so more lanes = fewer trips
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* p = BIG ? a.work : lds;        // 3*K*K
  __shared__ uint32_t sred[8];
  __shared__ float ssum[4];
  const int K = a.K, n = 3 * K * K, tid = threadIdx.x;
  // everything the step needs from global memory is requested up front (this kernel is one latency chain: one workgroup, ~8 barriers)
  constexpr int GR = BIG ? 1 : (3 * 63 * 63 + NTHR - 1) / NTHR;   // gradient values per thread at the largest PSF (BIG: read in place)
  float gk[GR];
  int frozen = 0;
  if (a.do_step) {
    frozen = *a.frozen;
    if (!BIG) {
#pragma unroll
      for (int r = 0; r < GR; ++r) { const int i = tid + r * NTHR; gk[r] = i < n ? a.gradk[i] : 0.f; }
    }
  }
  for (int i = tid; i < n; i += NTHR) p[i] = a.psf[i];
  if (tid < 8) sred[tid] = 0u;
  __syncthreads();
  if (a.do_step) {
    uint32_t kp = 0u, kg = 0u;
    if (BIG) {
      for (int i = tid; i < n; i += NTHR) {
        const uint32_t k1 = key_of(p[i]), k2 = key_of(__builtin_fabsf(a.gradk[i]));
        kp = kp > k1 ? kp : k1; kg = kg > k2 ? kg : k2;
      }
    } else {
#pragma unroll
      for (int r = 0; r < GR; ++r) {
        const int i = tid + r * NTHR;
        if (i < n) {
          const uint32_t k1 = key_of(p[i]), k2 = key_of(__builtin_fabsf(gk[r]));
          kp = kp > k1 ? kp : k1; kg = kg > k2 ? kg : k2;
        }
      }
    }
    kp = wave_max_u32(kp); kg = wave_max_u32(kg);
    if ((tid & 63) == 0) { atomicMax(&sred[0], kp); atomicMax(&sred[1], kg); }
    __syncthreads();
    const float maxp = ics_key2f(sred[0]), maxg = ics_key2f(sred[1]);
    const float dtpsf = __fdiv_rn(__fmul_rn(__fdiv_rn(a.step, (float)K), maxp), __fadd_rn(maxg, 1e-15f));
    if (tid == 0) a.scal[ICS_SC_DTPSF] = dtpsf;
    if (BIG) {
      for (int i = tid; i < n; i += NTHR) {
        p[i] = __fsub_rn(p[i], __fmul_rn(dtpsf, a.gradk[i]));
        if (!frozen) a.psf_caller[i] = p[i];
      }
    } else {
#pragma unroll
      for (int r = 0; r < GR; ++r) {
        const int i = tid + r * NTHR;
        if (i < n) {
          p[i] = __fsub_rn(p[i], __fmul_rn(dtpsf, gk[r]));
          if (!frozen) a.psf_caller[i] = p[i];
        }
      }
    }
    __syncthreads();
    if (a.correlation) {
      for (int i = tid; i < K * K; i += NTHR) {
        const float mval = __fdiv_rn(__fadd_rn(__fadd_rn(p[3*i], p[3*i+1]), p[3*i+2]), 3.0f);
        p[3*i] = mval; p[3*i+1] = mval; p[3*i+2] = mval;
      }
      __syncthreads();
    }
    for (int i = tid; i < n; i += NTHR) if (p[i] < 0.f) p[i] = 0.f;
    __syncthreads();
    if (tid < 3) {  // sequential float32 sum in the reference's order (i, j) -- pyx:58-64
      // (the adds are a dependent chain by definition; the LDS reads are not: 32 of them are requested before the adds that use
      //  them -- one read per add cost ~100 cycles per tap, 10 us of this kernel at 15x15 and 45 us at 31x31)
      float s = 0.f;
      const int KK = K * K;
      int i = 0;
      for (; i + 32 <= KK; i += 32) {
        float v[32];
#pragma unroll
        for (int k = 0; k < 32; ++k) v[k] = p[3 * (i + k) + tid];
#pragma unroll
        for (int k = 0; k < 32; ++k) s = __fadd_rn(s, v[k]);
      }
      for (; i < KK; ++i) s = __fadd_rn(s, p[3 * i + tid]);
      ssum[tid] = s;
    }
    __syncthreads();
    const bool detach = a.correlation != 0;
    for (int i = tid; i < n; i += NTHR) {
      p[i] = __fdiv_rn(p[i], ssum[i % 3]);
      a.psf[i] = p[i];
      if (!frozen && !detach) a.psf_caller[i] = p[i];
    }
    if (tid == 0 && detach) *a.frozen = 1;
    __syncthreads();
  }
  // pack the row-pair weights of the convolution kernels (ics_common.h, IcsConvArgs::w):
  //   W_corr[a][b][c] = psf[a][b][c] (A3),  W_conv[a][b][c] = psf[K-1-a][K-1-b][c] (A1, = psf_rotated, pyx:589)
  //   w[ap][(3b+c)*2 + h] = W[ap - h][b][c], ap = 0..K, W[-1] = W[K] = 0; row padding zeroed
  for (int i = tid; !BIG && i < (K + 1) * a.wrow; i += NTHR) {
    const int ap = i / a.wrow, rc = i - ap * a.wrow;
    float v1 = 0.f, v2 = 0.f;
    if (rc < 6 * K) {
      const int h = rc & 1, bc = rc >> 1, b = bc / 3, c = bc - 3 * b;
      const int ra = ap - h;
      if (ra >= 0 && ra < K) {
        v1 = p[(ra * K + b) * 3 + c];
        v2 = p[((K - 1 - ra) * K + (K - 1 - b)) * 3 + c];
      }
    }
    a.wcorr[i] = v1; a.wconv[i] = v2;
  }
  // Weight tables of the matrix-core convolution (ics_conv_mfma.hip; = its LDS image): every weight scaled by a power
  // of two (max -> [2^14, 2^15)) and split into fp16 hi + lo.  Row (c*K + a)*2 + s holds halves 8 .. of the zero-padded
  // kernel row Wp[idx] = W[a][idx - 15][c], i.e. the taps at local halves 7 .. K+6 and zeros around them.
  if (a.bt_conv && a.bt_corr) {
    uint32_t km = 0u;
    for (int i = tid; i < n; i += NTHR) { const uint32_t k1 = key_of(__builtin_fabsf(p[i])); km = km > k1 ? km : k1; }
    km = wave_max_u32(km);
    if ((tid & 63) == 0) atomicMax(&sred[2], km);
    __syncthreads();
    const float m = ics_key2f(sred[2]);
    const uint32_t e = (__float_as_uint(m) >> 23) & 0xFFu;
    uint32_t sb = 127u;
    if (m > 0.f && e != 255u) { sb = 268u - e; sb = sb > 240u ? 240u : sb; }
    const float s_w = __uint_as_float(sb << 23), inv_w = __uint_as_float((254u - sb) << 23);
    _Float16* tc = reinterpret_cast<_Float16*>(a.bt_conv);
    _Float16* tr = reinterpret_cast<_Float16*>(a.bt_corr);
    const int rh = ((2 * (K + 17) + 3) & ~3) / 2;      // halves per row (MCfg::WROWB / 2)
    const int nhalf = 3 * K * 2 * rh;
    for (int i = tid; i < nhalf; i += NTHR) {
      const int ent = i / rh, hh = i - ent * rh;
      const int sp = ent & 1, ca = ent >> 1, c = ca / K, ra = ca - c * K;
      const int b = hh - 7;
      float w1 = 0.f, w2 = 0.f;
      if (b >= 0 && b < K) {
        w1 = p[(ra * K + b) * 3 + c] * s_w;
        w2 = p[((K - 1 - ra) * K + (K - 1 - b)) * 3 + c] * s_w;
      }
      const _Float16 h1 = (_Float16)w1, h2 = (_Float16)w2;
      // the two split terms of a row are interleaved dword by dword (hi dword d at 2d, lo at 2d + 1): one 8-byte LDS read
      // fetches both
      const int o = ca * 2 * rh + 4 * (hh >> 1) + 2 * sp + (hh & 1);
      tr[o] = sp ? (_Float16)(w1 - (float)h1) : h1;
      tc[o] = sp ? (_Float16)(w2 - (float)h2) : h2;
    }
    if (tid == 0) {
      *reinterpret_cast<float*>(tc + nhalf) = inv_w;
      *reinterpret_cast<float*>(tr + nhalf) = inv_w;
    }
  }
}


// The same pass, lane-contiguous: a wave owns 256 consecutive floats of a row (1 KiB per frame and instruction instead
// of 64 x 16 B at a 48-byte stride, a third of the cache-line requests); three such segments per iteration keep 12+ loads
// in flight per lane.  A float's channel is (flat index) mod 3: with r = (first flat index of the lane) mod 3 the lane's
// element e has channel (r + e) mod 3 and pixel q + (r + e >= 3).  Loads are streaming (nt): nothing read here is read
// again before ~1 GB of other traffic, and default loads evicted what the next kernel re-reads (0.197 -> 0.173 ms).
// Same arithmetic, bit-identical to k_update.  TVK: 0 shipped mode, 1 active MM-TV, 2 PAM (kinds 2 and 3).
template <int TVK>
__global__ __launch_bounds__(256) void k_update_rows(IcsUpdateArgs a) {
  const IcsGeom& G = a.geo;
  float dt[3], dt2[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float maxu = ics_key2f(a.red[ICS_RED_MAXU + c]);
    const float maxg = ics_key2f(a.red[ICS_RED_MAXG + c]);
    dt[c] = __fdiv_rn(__fmul_rn(a.step, maxu), __fadd_rn(maxg, 1e-15f));
    if (TVK == 1)   // pyx:548: dt = step*(max image_k + 0)/(max|gradu_k| + 1e-15) with gradu = T
      dt2[c] = __fdiv_rn(__fmul_rn(a.step, ics_key2f(a.red[ICS_RED_MAXF + c])), __fadd_rn(ics_key2f(a.red[ICS_RED_MAXT + c]), 1e-15f));
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      a.scal[ICS_SC_DT + c] = dt[c]; a.scal[ICS_SC_MAXU + c] = maxu; a.scal[ICS_SC_MAXG + c] = maxg;
    }
  }
  uint32_t kmin = 0xFFFFFFFFu, kmax = 0u, knan = 0u;
  const float lambd = a.lambd;
  const int lane = threadIdx.x & 63;
  const int rowf = 3 * G.uN;                       // floats per row
  const int nwc = (rowf + 255) / 256;              // wave segments per row
  const long nitems = (long)G.uM * nwc;
  const long gw = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (long)gridDim.x * 4;
  constexpr int U = TVK == 0 ? ICS_UPDATE_U : (TVK == 1 ? ICS_UPDATE_U_TV : 2);
  for (long it0 = gw * U; it0 < nitems; it0 += nw * U) {
    f32x4 uq[U], tq[U], gq[U], fq[U], Tq[U];
    int ys[U], f0s[U];
#pragma unroll
    for (int s = 0; s < U; ++s) {
      long it = it0 + s; it = it < nitems ? it : nitems - 1;
      const int y = (int)(it / nwc), wc = (int)(it - (long)y * nwc);
      ys[s] = __builtin_amdgcn_readfirstlane(y);
      f0s[s] = 256 * __builtin_amdgcn_readfirstlane(wc) + 4 * lane;
      const ptrdiff_t o = (ptrdiff_t)ys[s] * G.pitch + f0s[s];
      const f32x4* pu = reinterpret_cast<const f32x4*>(a.u + o); const f32x4* pt = reinterpret_cast<const f32x4*>(a.ut + o);
      const f32x4* pg = reinterpret_cast<const f32x4*>(a.g + o); const f32x4* pf = reinterpret_cast<const f32x4*>(a.f + o);
      uq[s] = (ICS_UPDATE_NT & 1) ? __builtin_nontemporal_load(pu) : *pu;
      if (TVK != 2) tq[s] = (ICS_UPDATE_NT & 2) ? __builtin_nontemporal_load(pt) : *pt;
      gq[s] = (ICS_UPDATE_NT & 4) ? __builtin_nontemporal_load(pg) : *pg;
      if (TVK != 2) fq[s] = (ICS_UPDATE_NT & 8) ? __builtin_nontemporal_load(pf) : *pf;
      if (TVK == 1) Tq[s] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(a.tv + o));
    }
#pragma unroll
    for (int s = 0; s < U; ++s) {
      if (it0 + s >= nitems) break;                // wave-uniform
      const int y = ys[s], f0 = f0s[s];
      if (f0 >= rowf) continue;
      const int q = f0 / 3, r = f0 - 3 * q;
      const float dtr[3] = {r == 0 ? dt[0] : (r == 1 ? dt[1] : dt[2]), r == 0 ? dt[1] : (r == 1 ? dt[2] : dt[0]), r == 0 ? dt[2] : (r == 1 ? dt[0] : dt[1])};
      const float dt2r[3] = {r == 0 ? dt2[0] : (r == 1 ? dt2[1] : dt2[2]), r == 0 ? dt2[1] : (r == 1 ? dt2[2] : dt2[0]), r == 0 ? dt2[2] : (r == 1 ? dt2[0] : dt2[1])};
      const bool yin = (y >= G.pad) && (y < G.pad + G.M);
      const ptrdiff_t o = (ptrdiff_t)y * G.pitch + f0;
      float un4[4], fn4[4];
      bool in4[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int x = q + ((r + e) >= 3 ? 1 : 0);
        const bool inside = yin && (x >= G.pad) && (x < G.pad + G.N);
        in4[e] = inside;
        const float uv = uq[s][e], gv = gq[s][e];
        float g;
        if (TVK == 2)                                                            // PAM: the back-projection pass wrote G = T + lambd*gradu
          g = gv;
        else if (TVK == 1 && y >= 1 && y <= G.uM - 2 && x >= 1 && x <= G.uN - 2)   // pyx:517 (TV_ut_L1 != 0 and TV_u_L1 != 0)
          g = (float)(((double)Tq[s][e] + (double)__fmul_rn(lambd, gv)) + (double)__fsub_rn(uv, tq[s][e]) / 4.0);
        else
          g = __fadd_rn(__fmul_rn(lambd, gv), __fmul_rn(__fsub_rn(uv, tq[s][e]), 0.5f));
        float un = __fsub_rn(uv, __fmul_rn(dtr[e % 3], g));
        if (inside && TVK != 2) {   // (PAM has no DoF blend)
          float fv = fq[s][e];
          const float d = ics_dof_ratio(gv, fv);
          float D = __fmul_rn(d, d);
          if (!a.blind) D = __fdiv_rn(D, lambd);
          if (TVK == 1) {  // pyx:549: image -= dt*gradu/lambd, then the blend uses the updated image
            fv = __fsub_rn(fv, __fdiv_rn(__fmul_rn(dt2r[e % 3], Tq[s][e]), lambd));
            fn4[e] = fv;
          }
          un = __fadd_rn(__fmul_rn(__fsub_rn(1.0f, D), un), __fmul_rn(D, fv));
          if (a.want_dof) {
            if (D != D) knan = 1u;
            else { const uint32_t k = ics_f2key(D); kmin = kmin < k ? kmin : k; kmax = kmax > k ? kmax : k; }
          }
        }
        un4[e] = un;
      }
      if (TVK == 1) {   // the image step: one 16-byte store where the four floats are image pixels, single floats on the rim
        if (in4[0] && in4[3]) {
          const f32x4 w = {fn4[0], fn4[1], fn4[2], fn4[3]};
          *reinterpret_cast<f32x4*>(a.f_rw + o) = w;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (in4[e]) a.f_rw[o + e] = fn4[e];
        }
      }
      if (f0 + 3 < rowf) {
        const f32x4 w = {un4[0], un4[1], un4[2], un4[3]};
        if (ICS_UPDATE_NT & 16) __builtin_nontemporal_store(w, reinterpret_cast<f32x4*>(a.u_out + o));
        else *reinterpret_cast<f32x4*>(a.u_out + o) = w;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (f0 + e < rowf) a.u_out[o + e] = un4[e];
      }
    }
  }
  if (a.want_dof) {   // wave -> workgroup -> one atomic per workgroup and value, skipped where the running extremum already covers it
    __shared__ uint32_t shd[4][3];
    kmin = wave_min_u32(kmin); kmax = wave_max_u32(kmax); knan = wave_max_u32(knan);
    if ((threadIdx.x & 63) == 0) { shd[threadIdx.x >> 6][0] = kmin; shd[threadIdx.x >> 6][1] = kmax; shd[threadIdx.x >> 6][2] = knan; }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
      for (int w = 1; w < 4; ++w) { kmin = kmin < shd[w][0] ? kmin : shd[w][0]; kmax = kmax > shd[w][1] ? kmax : shd[w][1]; knan |= shd[w][2]; }
      if (kmin < a.dofkeys[0]) atomicMin(a.dofkeys + 0, kmin);
      if (kmax > a.dofkeys[1]) atomicMax(a.dofkeys + 1, kmax);
      if (knan) atomicOr(a.dofkeys + 2, 1u);
    }
  }
}

// ---- row bands: the two global quantities of an inner iteration restricted to the rows a band owns -------
// max |lambd*gradu + (u - ut)/2| and max u per channel (pyx:519,523-524, shipped regulariser) over u-frame rows [r0, r1)
__global__ __launch_bounds__(256) void k_band_reduce(const float* __restrict__ gr, const float* __restrict__ u, const float* __restrict__ ut,
                                                    IcsGeom G, float lambd, int r0, int r1, uint32_t* __restrict__ red) {
  __shared__ uint32_t sh[4 * 8];
  // four consecutive floats of a row per thread and step (rows start 16-byte aligned: the pitch is a multiple of 64 floats); what lies
  // between 3 uN and the next multiple of 4 is apron, read and left out.  (One float per thread took 1.6 ms at 12288^2.)
  typedef float f4 __attribute__((ext_vector_type(4)));
  const int rowf = 3 * G.uN, rowq = (rowf + 3) / 4;
  const long total = (long)(r1 - r0) * rowq;
  uint32_t kg[3] = {0u, 0u, 0u}, ku[3] = {0u, 0u, 0u};
  for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
    const int y = r0 + (int)(t / rowq), f = 4 * (int)(t - (long)(y - r0) * rowq);
    const ptrdiff_t o = (ptrdiff_t)y * G.pitch + f;
    const f4 uq = __builtin_nontemporal_load(reinterpret_cast<const f4*>(u + o)), tq = __builtin_nontemporal_load(reinterpret_cast<const f4*>(ut + o));
    const f4 gq = __builtin_nontemporal_load(reinterpret_cast<const f4*>(gr + o));
    int c = f % 3;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float uv = uq[e];
      const float g = __fadd_rn(__fmul_rn(lambd, gq[e]), __fmul_rn(__fsub_rn(uv, tq[e]), 0.5f));
      const uint32_t k1 = f + e < rowf ? key_of(__builtin_fabsf(g)) : 0u, k2 = f + e < rowf ? key_of(uv) : 0u;
#pragma unroll
      for (int cc = 0; cc < 3; ++cc) if (cc == c) { kg[cc] = kg[cc] > k1 ? kg[cc] : k1; ku[cc] = ku[cc] > k2 ? ku[cc] : k2; }
      c = c == 2 ? 0 : c + 1;
    }
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int c = 0; c < 3; ++c) { kg[c] = wave_max_u32(kg[c]); ku[c] = wave_max_u32(ku[c]); }
  if (lane == 0) {
#pragma unroll
    for (int c = 0; c < 3; ++c) { sh[wave * 8 + c] = kg[c]; sh[wave * 8 + 3 + c] = ku[c]; }
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    uint32_t m = sh[threadIdx.x];
    for (int w = 1; w < 4; ++w) m = m > sh[w * 8 + threadIdx.x] ? m : sh[w * 8 + threadIdx.x];
    atomicMax(red + (threadIdx.x < 3 ? ICS_RED_MAXG + threadIdx.x : ICS_RED_MAXU + (threadIdx.x - 3)), m);
  }
}
// residual rows outside image rows [i0, i1) := 0 (frame rows i + pad): only those rows are walked
__global__ __launch_bounds__(256) void k_band_mask_e(float* __restrict__ e, IcsGeom G, int i0, int i1) {
  const int rowf = 3 * G.N;
  const int nout = i0 + (G.M - i1);                 // rows [0, i0) and [i1, M)
  const long total = (long)nout * rowf;
  for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
    const int k = (int)(t / rowf), f = (int)(t - (long)k * rowf);
    const int i = k < i0 ? k : i1 + (k - i0);
    e[(ptrdiff_t)(i + G.pad) * G.pitch + 3 * G.pad + f] = 0.f;
  }
}

}  // namespace

hipError_t ics_launch_band_reduce(const float* gr, const float* u, const float* ut, const IcsGeom& g, float lambd, int r0, int r1, uint32_t* red, hipStream_t s) {
  hipLaunchKernelGGL(k_band_reduce, dim3(1024), dim3(256), 0, s, gr, u, ut, g, lambd, r0, r1, red);
  return hipGetLastError();
}
hipError_t ics_launch_band_mask_e(float* e, const IcsGeom& g, int i0, int i1, hipStream_t s) {
  if (i0 + (g.M - i1) <= 0) return hipSuccess;     // the band owns every row
  hipLaunchKernelGGL(k_band_mask_e, dim3(1024), dim3(256), 0, s, e, g, i0, i1);
  return hipGetLastError();
}

hipError_t ics_launch_tvterm(const IcsTvTermArgs& a, hipStream_t s) {
  const bool exact = ics_debug().pam_exact.load(std::memory_order_relaxed) != 0 && !a.planar;   // (the per-value cross-check form reads HWC frames)
  if (a.planar && a.kind < 2) return hipErrorInvalidValue;
  if (a.kind == 1 && exact) hipLaunchKernelGGL(k_tvterm<1>, dim3(1024), dim3(256), 0, s, a, 1);
  else if (a.kind == 1) {
    // as for the PAM kinds below: enough threads for ~4 waves per SIMD, at most ICS_TVMM_SEG rows per thread
    const long cols = (long)a.geo.tiles_x * 16;
    int seg = (int)(((long)a.geo.uM * cols) / 262144);
    seg = seg < 2 ? 2 : (seg > ICS_TVMM_SEG ? ICS_TVMM_SEG : seg);
    const long nthreads = (((long)a.geo.uM + seg - 1) / seg) * cols;
    long nblk = (nthreads + 255) / 256; nblk = nblk > 2048 ? 2048 : (nblk < 1 ? 1 : nblk);
    hipLaunchKernelGGL((k_tvterm<1, true>), dim3((unsigned)nblk), dim3(256), 0, s, a, seg);
  }
  else if (exact) {   // the per-value IEEE form (pam_term): kept as the cross-check
    if (a.kind == 2) hipLaunchKernelGGL(k_tvterm<2>, dim3(1024), dim3(256), 0, s, a, 16);
    else hipLaunchKernelGGL(k_tvterm<3>, dim3(1024), dim3(256), 0, s, a, 16);
  }
  else {
    // a thread = 4 pixels x `seg` rows; keep >= ~4 waves per SIMD busy: a 2048^2 frame walked 16 rows at a time is one wave per SIMD,
    // each a serial chain of dependent loads (measured 0.14 ms there against 0.11 ms for the 4096^2 frame)
    const long cols = (long)a.geo.tiles_x * 16;
    int seg = (int)(((long)a.geo.uM * cols) / 262144);
    seg = seg < 2 ? 2 : (seg > 16 ? 16 : seg);
    const long nthreads = (((long)a.geo.uM + seg - 1) / seg) * cols;
    long nblk = (nthreads + 255) / 256; nblk = nblk > 2048 ? 2048 : (nblk < 1 ? 1 : nblk);
    if (a.planar) {
      if (a.kind == 2) hipLaunchKernelGGL((k_tvterm_pam<false, true>), dim3((unsigned)nblk), dim3(256), 0, s, a, seg);
      else hipLaunchKernelGGL((k_tvterm_pam<true, true>), dim3((unsigned)nblk), dim3(256), 0, s, a, seg);
    }
    else if (a.kind == 2) hipLaunchKernelGGL((k_tvterm_pam<false, false>), dim3((unsigned)nblk), dim3(256), 0, s, a, seg);
    else hipLaunchKernelGGL((k_tvterm_pam<true, false>), dim3((unsigned)nblk), dim3(256), 0, s, a, seg);
  }
  return hipGetLastError();
}

hipError_t ics_launch_update(const IcsUpdateArgs& a, hipStream_t s) {
  const long total = (long)a.geo.uM * a.geo.tiles_x * 16;
  long blocks = (total + 255) / 256;
  // Few long-lived workgroups stream best here.  k_update_rows at 4096^2 (4 reads + 1 write, 1.0 GB), segments per loop
  // iteration x workgroups per CU: 3x2 0.173 ms, 4x2 0.167, 4x3 0.164, 4x4 0.170, 6x2 0.164, 6x4 0.180 (scripts/sweep_update.sh).
  // ... and at smaller frames fewer still: 2048^2 0.0568 / 0.0526 / 0.0565 ms and 3072^2 0.119 / 0.099 / 0.102 ms with 1 / 2 / 3 per CU,
  // 1024^2 0.0252 / 0.0265 / 0.0299
  const long px = (long)a.geo.uM * a.geo.uN;
  const int per_cu_env = ics_debug().update_wg_per_cu.load(std::memory_order_relaxed);
  const int per_cu = per_cu_env > 0 ? per_cu_env : (px >= 12000000L ? 3 : (px >= 1500000L ? 2 : 1));
  const long cap = (long)ics_device_cus(ics_current_device()) * (per_cu > 0 ? per_cu : 3);
  if (blocks > cap) blocks = cap;
  const int rows_kernel = ics_debug().update_kernel.load(std::memory_order_relaxed);   // 0: the pixel-group kernel everywhere
  const int tvk = (a.tv && a.tv_kind) ? (a.tv_kind >= 2 ? 2 : 1) : 0;
  if (rows_kernel && tvk == 0) hipLaunchKernelGGL(k_update_rows<0>, dim3((unsigned)cap), dim3(256), 0, s, a);
  else if (rows_kernel && tvk == 1) hipLaunchKernelGGL(k_update_rows<1>, dim3((unsigned)cap), dim3(256), 0, s, a);
  else if (rows_kernel && tvk == 2) hipLaunchKernelGGL(k_update_rows<2>, dim3((unsigned)cap), dim3(256), 0, s, a);
  else hipLaunchKernelGGL(k_update, dim3((unsigned)blocks), dim3(256), 0, s, a);
  return hipGetLastError();
}

int ics_gradk_blocks(const IcsGeom& g, int cus) {
  const int nb = (g.K + 15) / 16;
  const int tiles = g.tiles_x * g.tiles_y * 2;
  int blocks = cus * (nb <= 2 ? 2 : 1);   // two persistent workgroups per CU where the matrix-core kernel exists
  if (const int m = ics_debug().max_wgs.load(std::memory_order_relaxed); m > 0 && blocks > m) blocks = m;   // test hook (ics_debug.h)
  return blocks < tiles ? blocks : tiles;
}

template <int NB>
static hipError_t launch_gradk(const IcsGradkArgs& a, int nblocks, hipStream_t s) {
  constexpr int NW = ICS_GRADK_WAVES;
  using C = GradkCfg<NB, NW>;
  static std::atomic<bool> configured[ICS_MAX_DEVICES];  // per device: the dynamic-LDS attribute is a per-device function property
  const int dev = ics_current_device();
  auto kern = k_gradk<NB, NW>;
  if (hipError_t e = ics_configure_lds(configured, dev, kern, C::LDS_BYTES); e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3(nblocks), dim3(C::NTH), C::LDS_BYTES, s, a);
  return hipGetLastError();
}

hipError_t ics_launch_gradk(const IcsGradkArgs& a, int nblocks, hipStream_t s) {
  const int nb = (a.geo.K + 15) / 16;
  if (nb == 1) return launch_gradk<1>(a, nblocks, s);
  if (nb == 2) return launch_gradk<2>(a, nblocks, s);
  if (nb == 3) return launch_gradk<3>(a, nblocks, s);
  if (nb == 4) return launch_gradk<4>(a, nblocks, s);
  return hipErrorInvalidValue;
}

hipError_t ics_launch_gradk_reduce(const float* partial, int nblocks, float* gradk, const IcsGeom& g, hipStream_t s) {
  const int nt = 16 * ((g.K + 15) / 16);
  hipLaunchKernelGGL(k_gradk_reduce, dim3((3 * nt * nt + 63) / 64), dim3(1024), 0, s, partial, nblocks, gradk, nt, g.K, g.K, g.K, 0, 0);
  return hipGetLastError();
}

hipError_t ics_launch_gradk_reduce_block(const float* partial, int nblocks, float* gradk, int nt, int La, int Lb, int Kf, int a0, int b0, hipStream_t s) {
  hipLaunchKernelGGL(k_gradk_reduce, dim3((3 * nt * nt + 63) / 64), dim3(1024), 0, s, partial, nblocks, gradk, nt, La, Lb, Kf, a0, b0);
  return hipGetLastError();
}

namespace {
__global__ __launch_bounds__(256) void k_zero_many(IcsZeroArgs a) {
  const unsigned long long total = a.end16[a.count - 1];
  const unsigned long long stride = (unsigned long long)gridDim.x * 256;
  int e = 0;
  for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
    while (i >= a.end16[e]) ++e;                                   // (i only grows)
    const unsigned long long first = e ? a.end16[e - 1] : 0ull;
    reinterpret_cast<uint4*>(a.p[e])[i - first] = make_uint4(0u, 0u, 0u, 0u);
  }
}
}  // namespace

namespace {
__global__ __launch_bounds__(256) void k_run_reset(IcsRunResetArgs a) {
  const int t = threadIdx.x;
  if (t < 4) a.flags[t] = 0;
  if (t < 16) a.sched[t] = 0u;
  if (t < 8) a.dacc[t] = 0.0;
  if (t < 2) a.ukey[t] = 0u;
  if (t < 8) a.dofkeys[t] = (t & 3) == 0 ? 0xFFFFFFFFu : 0u;
  for (int i = t; i < a.nred; i += 256) a.red[i] = 0u;
}
}  // namespace
hipError_t ics_launch_run_reset(const IcsRunResetArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(k_run_reset, dim3(1), dim3(256), 0, s, a);
  return hipGetLastError();
}

hipError_t ics_launch_zero_many(const IcsZeroArgs& a, hipStream_t s) {
  if (a.count <= 0) return hipSuccess;
  const unsigned long long total = a.end16[a.count - 1];
  if (!total) return hipSuccess;
  unsigned long long blocks = (total + 256 * 8 - 1) / (256 * 8);
  const unsigned long long cap = (unsigned long long)ics_device_cus(ics_current_device()) * 8;
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(k_zero_many, dim3((unsigned)blocks), dim3(256), 0, s, a);
  return hipGetLastError();
}

hipError_t ics_launch_psf(const IcsPsfArgs& a, hipStream_t s) {
  if (a.K > 63) {
    if (!a.work) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_psf<true>, dim3(1), dim3(ICS_PSF_THREADS), 0, s, a);
  } else {
    hipLaunchKernelGGL(k_psf<false>, dim3(1), dim3(ICS_PSF_THREADS), (size_t)3 * a.K * a.K * sizeof(float), s, a);
  }
  return hipGetLastError();
}
