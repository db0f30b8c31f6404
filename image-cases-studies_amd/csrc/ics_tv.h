// ics_tv.h -- the 3x3 TV stencil of lib/deconvolution.pyx:137-239 as a device function, shared by the
// standalone operator (k_tv, ics_filters.hip) and the active MM-TV mode (k_tvterm, ics_kernels.hip).
// Arithmetic as the COMPILED reference evaluates it, pinned by tests/golden/tv.npz (oracle/make_golden_tv.py): Cython writes the
// literal of `-2 * u[i, j, k]` as the C double -2.0, so the second-order sums are double expressions rounded once
// (lib/deconvolution.c:4176); the numerator of the diagonal terms is rounded to float before the float division by dxdy;
// `adjust` is a double expression stored in a float (:4031); first-order differences stay float.  Everything else:
// separately rounded float32 operations in the reference's order (build with -ffp-contract=off).
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
  const float adjust = (norm == 1) ? (float)(4.0 * (1.0 + 1.0 / (double)dxdy))   // pyx:149-152, evaluated in double
                                   : (float)(2.0 * (1.0 + (double)dxdy));
  float d, r;
  if (order == 2) {  // pyx:156-189
    const double m2c = -2.0 * (double)c;   // exact
    const float udx = (float)__dadd_rn(__dadd_rn(m2c, (double)up), (double)dn);
    const float udy = (float)__dadd_rn(__dadd_rn(m2c, (double)lf), (double)rt);
    const float udxdy = __fdiv_rn((float)__dadd_rn(__dadd_rn(m2c, (double)ul), (double)dr), dxdy);
    const float udydx = __fdiv_rn((float)__dadd_rn(__dadd_rn(m2c, (double)ur), (double)dl), dxdy);
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
