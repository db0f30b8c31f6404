// ics_planar.hip -- channel-planar mirrors of the job's frames (ics_common.h: ics_ppitch, ics_plane_floats) for the FFT-tile pipeline
// (ics_conv_fft.hip; ics_api.hip "planar pipeline").
//
// The overlap-save FFT convolution works on one channel of a tile pair at a time; from HWC frames every access touches 4 of each 12 bytes
// and a work unit pulls three times its useful cache lines through the CU's miss path (measured: 30 k of a unit's 70 k shader clocks in the
// window loads alone).  During a run that uses it, the frames therefore live as three planes each, and the kernels between the
// convolutions run on the planes as well:
//   k_hwc_to_planar / k_planar_to_hwc   run boundaries (whole frames), and the stop-test window (A18 / A19 read HWC frames)
//   k_update_planar                     A5 + A6 + A8 + A10 (lib/deconvolution.pyx:499-552) -- the arithmetic of k_update_rows, operation for
//                                       operation (every operation rounded separately), so the results are bit-identical to the HWC pass
// The PSF gradient reads planes through a template flag of its own kernel (ics_gradk_mfma.hip).
#include "ics_kernels.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

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

// One thread = 4 consecutive pixels of a buffer row: 3 dwordx4 of HWC <-> one dwordx4 per plane.  Rows [r0, r1) and pixel groups [q0, q1)
// in BUFFER coordinates (row 0 = first allocated row, pixel 0 = first pixel of a row; ax is a multiple of 4, so frame pixel 0 starts a group).
template <bool TO_PLANAR>
__global__ __launch_bounds__(256) void k_planar_convert(const float* __restrict__ src, float* __restrict__ dst, int pitch, int ppitch, size_t plane,
                                                        int r0, int r1, int q0, int q1) {
  const int nq = q1 - q0;
  const long total = (long)(r1 - r0) * nq;
  for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
    const int row = r0 + (int)(t / nq), q = q0 + (int)(t - (long)(row - r0) * nq);
    const size_t oh = (size_t)row * pitch + 12 * (size_t)q, op = (size_t)row * ppitch + 4 * (size_t)q;
    const bool in_hwc = 12 * q + 11 < pitch;       // (the last groups of a plane row can lie beyond the HWC row: zeros)
    if (TO_PLANAR) {
      float v[12];
#pragma unroll
      for (int h = 0; h < 3; ++h) {
        const f32x4 a = in_hwc ? *reinterpret_cast<const f32x4*>(src + oh + 4 * h) : (f32x4){0.f, 0.f, 0.f, 0.f};
        v[4 * h] = a.x; v[4 * h + 1] = a.y; v[4 * h + 2] = a.z; v[4 * h + 3] = a.w;
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) *reinterpret_cast<f32x4*>(dst + c * plane + op) = (f32x4){v[c], v[3 + c], v[6 + c], v[9 + c]};
    } else if (in_hwc) {
      float v[12];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(src + c * plane + op);
        v[c] = a.x; v[3 + c] = a.y; v[6 + c] = a.z; v[9 + c] = a.w;
      }
#pragma unroll
      for (int h = 0; h < 3; ++h) *reinterpret_cast<f32x4*>(dst + oh + 4 * h) = (f32x4){v[4 * h], v[4 * h + 1], v[4 * h + 2], v[4 * h + 3]};
    }
  }
}

// A5 + A6 + A8 + A10 on planes: one thread = 4 consecutive pixels of one channel row (dwordx4 per operand).
//   g   = lambd*gradu + (u - ut)/2.                       (pyx:519)
//   dt  = step*max u_c / (max|g_c| + 1e-15)               (pyx:524)
//   u  -= dt*g; interior: D = ((gradu - f)/(gradu + f))^2 [/ lambd]; u = (1 - D) u + D f    (pyx:499-502, 531, 552)
__global__ __launch_bounds__(256) void k_update_planar(IcsUpdateArgs a, int ppitch, size_t plane) {
  const IcsGeom& G = a.geo;
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
  uint32_t kmin = 0xFFFFFFFFu, kmax = 0u, knan = 0u;
  const float lambd = a.lambd;
  const int nq = (G.uN + 3) / 4;
  const long per_plane = (long)G.uM * nq, total = 3 * per_plane;
  for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
    const int c = (int)(t / per_plane);
    const long r = t - (long)c * per_plane;
    const int y = (int)(r / nq), x0 = 4 * (int)(r - (long)y * nq);
    const size_t o = c * plane + (size_t)y * ppitch + x0;       // (frame origin: a.u etc. point at it)
    const f32x4 uq = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(a.u + o));
    const f32x4 tq = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(a.ut + o));
    const f32x4 gq = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(a.g + o));
    const f32x4 fq = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(a.f + o));
    const float dtc = c == 0 ? dt[0] : (c == 1 ? dt[1] : dt[2]);
    const bool yin = (y >= G.pad) && (y < G.pad + G.M);
    float un4[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int x = x0 + e;
      const bool inside = yin && (x >= G.pad) && (x < G.pad + G.N);
      const float uv = uq[e], gv = gq[e];
      // (PAM kinds: the back-projection wrote G = T + lambd gradu; no DoF blend -- k_update_rows, ics_kernels.hip)
      const float g = a.tv_kind >= 2 ? gv : __fadd_rn(__fmul_rn(lambd, gv), __fmul_rn(__fsub_rn(uv, tq[e]), 0.5f));
      float un = __fsub_rn(uv, __fmul_rn(dtc, g));
      if (inside && a.tv_kind < 2) {
        const float fv = fq[e];
        const float d = ics_dof_ratio(gv, fv);
        float D = __fmul_rn(d, d);
        if (!a.blind) D = __fdiv_rn(D, lambd);
        un = __fadd_rn(__fmul_rn(__fsub_rn(1.0f, D), un), __fmul_rn(D, fv));
        if (a.want_dof) {
          if (D != D) knan = 1u;
          else { const uint32_t k = ics_f2key(D); kmin = kmin < k ? kmin : k; kmax = kmax > k ? kmax : k; }
        }
      }
      un4[e] = un;
    }
    if (x0 + 3 < G.uN) *reinterpret_cast<f32x4*>(a.u_out + o) = (f32x4){un4[0], un4[1], un4[2], un4[3]};
    else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (x0 + e < G.uN) a.u_out[o + e] = un4[e];
    }
  }
  if (a.want_dof) {
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

}  // namespace

// `hwc` / `planar`: buffer STARTS (not origins).  Rows [y0, y1) and pixels [x0, x1) in u-frame coordinates, widened to whole 4-pixel groups
// and clipped to the buffer; whole = every allocated row and pixel (aprons included).
hipError_t ics_launch_planar_convert(bool to_planar, const float* src, float* dst, const IcsGeom& g, bool whole, int y0, int y1, int x0, int x1, hipStream_t s) {
  const int pp = ics_ppitch(g);
  int r0 = 0, r1 = g.rows, q0 = 0, q1 = pp / 4;
  if (!whole) {
    r0 = g.ay + y0; r1 = g.ay + y1;
    r0 = r0 < 0 ? 0 : r0; r1 = r1 > g.rows ? g.rows : r1;
    q0 = (g.ax + x0) / 4; q1 = (g.ax + x1 + 3) / 4;
    q0 = q0 < 0 ? 0 : q0; q1 = q1 > pp / 4 ? pp / 4 : q1;
  }
  if (r1 <= r0 || q1 <= q0) return hipSuccess;
  const long total = (long)(r1 - r0) * (q1 - q0);
  long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (to_planar) hipLaunchKernelGGL(k_planar_convert<true>, dim3((unsigned)blocks), dim3(256), 0, s, src, dst, g.pitch, pp, ics_plane_floats(g), r0, r1, q0, q1);
  else hipLaunchKernelGGL(k_planar_convert<false>, dim3((unsigned)blocks), dim3(256), 0, s, src, dst, g.pitch, pp, ics_plane_floats(g), r0, r1, q0, q1);
  return hipGetLastError();
}

// the update pass on planar mirrors: the frame pointers of `a` are ORIGINS of planar buffers (plane 0)
hipError_t ics_launch_update_planar(const IcsUpdateArgs& a, hipStream_t s) {
  const long px = (long)a.geo.uM * a.geo.uN;
  const int per_cu_env = ics_debug().update_wg_per_cu.load(std::memory_order_relaxed);
  const int per_cu = per_cu_env > 0 ? per_cu_env : (px >= 12000000L ? 3 : (px >= 1500000L ? 2 : 1));
  long blocks = (3 * (long)a.geo.uM * ((a.geo.uN + 3) / 4) + 255) / 256;
  const long cap = (long)ics_device_cus(ics_current_device()) * per_cu * 2;
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(k_update_planar, dim3((unsigned)blocks), dim3(256), 0, s, a, ics_ppitch(a.geo), ics_plane_floats(a.geo));
  return hipGetLastError();
}
