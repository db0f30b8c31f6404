// ics_image_acc.h -- the blurry image in ACCUMULATOR ORDER: a read-only copy of `image` laid out the way the matrix-core
// kernels hold their results, so that the image operand of the residual  e = convolve(u, psf) - image
// (lib/deconvolution.pyx:482-488 and :561-565) arrives as one 16-byte load per lane.
//
// A wave of k_conv_mfma<K, 0, RS> / k_synth_gradk<K> owns the 16-column block `cb` of a (16 RS) x 64-pixel tile; after the
// matrix phase lane (li = lane & 15, lg = lane >> 4) holds, per channel and accumulator set t < RS, the four values
//     r = 0 .. 3  <->  pixel ( y0 + t + 4 RS lg + RS r ,  x0 + 16 cb + li )            (tiles start at the image origin).
// In the HWC frame those are 4 x RS x 3 single floats 12 bytes apart from the neighbouring lane's: the epilogue of the fused
// A11 + A13 kernel spent 0.037 of its 0.29 ms on 32 such requests per lane and tile (round-2 verdict).  Here the same values are
// stored as
//     acc[ ((((tile * 4 + cb) * 3 + ch) * RS + t) * 64 + lane) * 4 + r ]
// i.e. one aligned float4 per lane, 1 KiB contiguous per wave instruction, 3 RS loads per tile instead of 4 RS (dwordx3) or 8 RS.
// `image` is constant during a run in the shipped loop (pyx:545-549 subtract exactly 0; only tv_mode 1 writes it): the copy is
// made once per upload (ics_api.hip, ensure_image_acc) and costs one frame read + one frame write.
#pragma once
#include "ics_common.h"

static inline size_t ics_image_acc_floats(const IcsGeom& g, int RS) {
  const size_t tiles = (size_t)((g.N + 63) / 64) * ((g.M + 16 * RS - 1) / (16 * RS));
  return tiles * 4 * 3 * RS * 64 * 4;
}

template <int RS>
__global__ __launch_bounds__(256) void k_image_acc(const float* __restrict__ f /* image frame origin (u-frame coordinates) */, IcsGeom g,
                                                   float* __restrict__ out) {
  const int tpr = (g.N + 63) / 64, ntiles = tpr * ((g.M + 16 * RS - 1) / (16 * RS));
  const long nvec = (long)ntiles * 4 * 3 * RS * 64;
  for (long v = (long)blockIdx.x * 256 + threadIdx.x; v < nvec; v += (long)gridDim.x * 256) {
    const int lane = (int)(v & 63);
    long w = v >> 6;
    const int t = (int)(w % RS); w /= RS;
    const int ch = (int)(w % 3); w /= 3;
    const int cb = (int)(w & 3);
    const int tile = (int)(w >> 2);
    const int tyi = tile / tpr, txi = tile - tyi * tpr;
    const int li = lane & 15, lg = lane >> 4;
    const int x = txi * 64 + 16 * cb + li;                       // image coordinates
    float4 o;
    float* po = reinterpret_cast<float*>(&o);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int y = tyi * 16 * RS + t + 4 * RS * lg + RS * r;
      po[r] = (y < g.M && x < g.N) ? f[(size_t)(y + g.pad) * g.pitch + 3 * (x + g.pad) + ch] : 0.f;
    }
    reinterpret_cast<float4*>(out)[v] = o;
  }
}

static inline hipError_t ics_launch_image_acc(const float* f_origin, const IcsGeom& g, int RS, float* out, hipStream_t s) {
  if (RS == 2) hipLaunchKernelGGL(k_image_acc<2>, dim3(2048), dim3(256), 0, s, f_origin, g, out);
  else if (RS == 4) hipLaunchKernelGGL(k_image_acc<4>, dim3(2048), dim3(256), 0, s, f_origin, g, out);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}
