// ics_tv.h -- the 3x3 TV stencil of lib/deconvolution.pyx:137-239 as a device function, shared by the
// standalone operator (k_tv, ics_filters.hip) and the active MM-TV mode (k_tvterm, ics_kernels.hip).
// Separately rounded float32 operations in the reference's order (build with -ffp-contract=off).
#pragma once
#include <hip/hip_runtime.h>

struct IcsTvOut { float out, div; };

__device__ __forceinline__ float ics_tv_norm(float x, float y, float eps, int norm) {
  // pyx:129-134: norm_L1 = |x|+|y|+eps ; norm_L2 = powf(x^2 + y^2 + eps^2, 0.5)
  if (norm == 1) return __fadd_rn(__fadd_rn(__builtin_fabsf(x), __builtin_fabsf(y)), eps);
  return __fsqrt_rn(__fadd_rn(__fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y)), __fmul_rn(eps, eps)));
}

// c = u[i,j]; up/dn = u[i-1,j]/u[i+1,j]; lf/rt = u[i,j-1]/u[i,j+1]; ul/dr = u[i-1,j-1]/u[i+1,j+1];
// ur/dl = u[i-1,j+1]/u[i+1,j-1]
__device__ __forceinline__ IcsTvOut ics_tv_point(float c, float up, float dn, float lf, float rt, float ul, float dr, float ur,
                                                 float dl, float eps, int order, int norm) {
  const float dxdy = 1.41421354f;  // powf(2, 0.5) as a float (pyx:146)
  const float adjust = (norm == 1) ? __fmul_rn(4.0f, __fadd_rn(1.0f, __fdiv_rn(1.0f, dxdy)))   // pyx:149-152
                                   : __fmul_rn(2.0f, __fadd_rn(1.0f, dxdy));
  float d, r;
  if (order == 2) {  // pyx:156-189
    const float m2c = __fmul_rn(-2.0f, c);
    const float udx = __fadd_rn(__fadd_rn(m2c, up), dn);
    const float udy = __fadd_rn(__fadd_rn(m2c, lf), rt);
    const float udxdy = __fdiv_rn(__fadd_rn(__fadd_rn(m2c, ul), dr), dxdy);
    const float udydx = __fdiv_rn(__fadd_rn(__fadd_rn(m2c, ur), dl), dxdy);
    d = __fsub_rn(__fsub_rn(__fsub_rn(-udx, udy), udxdy), udydx);
    r = __fadd_rn(ics_tv_norm(udx, udy, eps, norm), ics_tv_norm(udxdy, udydx, eps, norm));
  } else {           // pyx:191-237
    const float udx_b = __fsub_rn(c, up), udy_b = __fsub_rn(c, lf);
    const float udx_f = __fadd_rn(-c, dn), udy_f = __fadd_rn(-c, rt);
    const float udxdy_b = __fdiv_rn(__fsub_rn(c, ul), dxdy), udydx_b = __fdiv_rn(__fsub_rn(c, ur), dxdy);
    const float udydx_f = __fdiv_rn(__fadd_rn(-c, dl), dxdy), udxdy_f = __fdiv_rn(__fadd_rn(-c, dr), dxdy);
    d = __fadd_rn(udx_b, udy_b); d = __fsub_rn(d, udx_f); d = __fsub_rn(d, udy_f);
    d = __fadd_rn(d, udxdy_b); d = __fadd_rn(d, udydx_b); d = __fsub_rn(d, udxdy_f); d = __fsub_rn(d, udydx_f);
    r = __fadd_rn(__fadd_rn(__fadd_rn(ics_tv_norm(udx_b, udy_b, eps, norm), ics_tv_norm(udx_f, udy_f, eps, norm)),
                            ics_tv_norm(udxdy_b, udydx_b, eps, norm)), ics_tv_norm(udxdy_f, udydx_f, eps, norm));
  }
  IcsTvOut o;
  o.div = __fdiv_rn(d, adjust);
  o.out = __fdiv_rn(r, adjust);
  return o;
}
