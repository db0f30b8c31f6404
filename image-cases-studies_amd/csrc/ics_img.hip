// ics_img.hip -- element-wise / copy kernels of the device-resident image objects (ics_img_*, include/ics_hip.h):
// what deconvolve.py does to its frames between two richardson_lucy_MM calls (pad_image :24-37, gamma :100-103 and
// :346-352, the float32 <-> float64 conversions around the bicubic resize :245-249).  H x W x 3 float32, HWC, contiguous.
#include "ics_kernels.h"

namespace {

// np.pad(mode="edge") on the two spatial axes (deconvolve.py:24-37); one output row per blockIdx.y
__global__ __launch_bounds__(256) void k_img_pad_edge(const float* __restrict__ in, int H, int W, float* __restrict__ out, int top, int left,
                                                      int OH, int OW) {
  int y = (int)blockIdx.y - top;
  y = y < 0 ? 0 : (y > H - 1 ? H - 1 : y);
  const int rowl = 3 * OW;
  for (int xc = blockIdx.x * 256 + threadIdx.x; xc < rowl; xc += gridDim.x * 256) {
    const int xq = xc / 3, c = xc - 3 * xq;
    int x = xq - left;
    x = x < 0 ? 0 : (x > W - 1 ? W - 1 : x);
    out[(long)blockIdx.y * rowl + xc] = in[((long)y * W + x) * 3 + c];
  }
}

// y = powf(clip01?(x / div), exponent) * mul     (every step rounded to float32 like the numpy expressions)
__global__ __launch_bounds__(256) void k_img_gamma(float* __restrict__ a, long n, float div, float exponent, float mul, int clip01) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    float x = __fdiv_rn(a[i], div);
    if (clip01) x = x < 0.f ? 0.f : (x > 1.f ? 1.f : x);   // np.clip: NaN stays NaN
    a[i] = __fmul_rn(powf(x, exponent), mul);
  }
}

__global__ __launch_bounds__(256) void k_f32_to_f64(const float* __restrict__ in, double* __restrict__ out, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) out[i] = (double)in[i];
}
__global__ __launch_bounds__(256) void k_f64_to_f32(const double* __restrict__ in, float* __restrict__ out, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) out[i] = (float)in[i];
}

template <typename T>
__global__ __launch_bounds__(256) void k_int_to_f32(const T* __restrict__ in, float* __restrict__ out, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) out[i] = (float)in[i];
}

inline unsigned grid_for(long n) { long b = (n + 255) / 256; return (unsigned)(b > 65536 ? 65536 : (b < 1 ? 1 : b)); }

}  // namespace

hipError_t ics_launch_img_pad_edge(const float* in, int H, int W, float* out, int top, int bottom, int left, int right, hipStream_t s) {
  const int OH = H + top + bottom, OW = W + left + right;
  const unsigned gx = (unsigned)((3 * OW + 255) / 256);
  hipLaunchKernelGGL(k_img_pad_edge, dim3(gx > 64 ? 64 : gx, (unsigned)OH), dim3(256), 0, s, in, H, W, out, top, left, OH, OW);
  return hipGetLastError();
}
hipError_t ics_launch_img_gamma(float* a, long n, float div, float exponent, float mul, int clip01, hipStream_t s) {
  hipLaunchKernelGGL(k_img_gamma, dim3(grid_for(n)), dim3(256), 0, s, a, n, div, exponent, mul, clip01);
  return hipGetLastError();
}
hipError_t ics_launch_f32_to_f64(const float* in, double* out, long n, hipStream_t s) {
  hipLaunchKernelGGL(k_f32_to_f64, dim3(grid_for(n)), dim3(256), 0, s, in, out, n);
  return hipGetLastError();
}
hipError_t ics_launch_f64_to_f32(const double* in, float* out, long n, hipStream_t s) {
  hipLaunchKernelGGL(k_f64_to_f32, dim3(grid_for(n)), dim3(256), 0, s, in, out, n);
  return hipGetLastError();
}
hipError_t ics_launch_int_to_f32(const void* in, int bytes_per_value, float* out, long n, hipStream_t s) {
  if (bytes_per_value == 1) hipLaunchKernelGGL(k_int_to_f32<unsigned char>, dim3(grid_for(n)), dim3(256), 0, s, (const unsigned char*)in, out, n);
  else if (bytes_per_value == 2) hipLaunchKernelGGL(k_int_to_f32<unsigned short>, dim3(grid_for(n)), dim3(256), 0, s, (const unsigned short*)in, out, n);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}
