// pt_display.hip -- the display packer: Denoiser::Denoise -> denoise_kernel (include/Denoiser.h:29-52,
// src/denoise.cu:9-29) behind pt_display_pack (include/ptcore.h).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pt_internal.h"

#pragma clang fp contract(off)

namespace pt {
// denoise_kernel: src/denoise.cu:9-29.  One lane per pixel, lanes along columns (the reference maps
// adjacent threads to adjacent rows); 12 B read + 12 B written per pixel: HBM-bound and tiny.
__global__ void __launch_bounds__(256) display_pack_kernel(const float* __restrict__ in, float* __restrict__ out, int width,
                                                           uint32_t pixels) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= pixels) return;
  const uint32_t row = i / (uint32_t)width, col = i % (uint32_t)width;
  const float* px = in + (size_t)i * 14;
  uint32_t packed = 1u << 24;  // uchar4 {r, g, b, 1}, little endian
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const float v = fminf(fmaxf(px[k], 0.0f), 1.0f);              // :18-20
    packed |= (uint32_t)(unsigned char)((double)v * 255.0) << (8 * k);  // :23
  }
  float* o = out + (size_t)i * 3;
  o[0] = (float)col;                     // :26
  o[1] = (float)(width - (int)row);      // :27
  o[2] = __uint_as_float(packed);        // :28
}
}  // namespace pt

#define PT_HIPD(call)                                                                             \
  do {                                                                                            \
    hipError_t e_ = (call);                                                                       \
    if (e_ != hipSuccess) return pt_fail(PT_EHIP, "%s: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

extern "C" int pt_display_pack(const float* d_buffer, int width, int height, float* d_vertices, void* hip_stream) {
  if (width <= 0 || height <= 0 || !d_buffer || !d_vertices) return pt_fail(PT_EINVAL, "pt_display_pack: bad arguments");
  const uint64_t pixels = (uint64_t)width * (uint64_t)height;
  if (pixels > 0xFFFFFFFFull) return pt_fail(PT_EINVAL, "pt_display_pack: image too large");
  hipLaunchKernelGGL(pt::display_pack_kernel, dim3((unsigned)((pixels + 255) / 256)), dim3(256), 0, (hipStream_t)hip_stream,
                     d_buffer, d_vertices, width, (uint32_t)pixels);
  PT_HIPD(hipGetLastError());
  return PT_OK;
}
