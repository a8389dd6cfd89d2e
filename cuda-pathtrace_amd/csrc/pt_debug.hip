// pt_debug.hip -- LAB LIBRARY ONLY (libptcore_lab.so, include/ptcore_lab.h): diagnostic entry points that evaluate the
// device-side scalar building blocks elementwise, and compare two of them over a range of
// float bit patterns.  Used by the tests to (1) check the device definitions of sin/cos,
// sqrt, 1/sqrt against the CPU oracle on dense samples and (2) PROVE by exhaustion that the
// cheap sequences of pt_device.h equal the literal expressions for every float.
#include "pt_device.h"
#include "pt_internal.h"
#include "pt_kernel.h"
#include "../../include/ptcore_lab.h"

#pragma clang fp contract(off)

namespace pt {

__device__ __forceinline__ float eval_unary(int fn, float x) {
  float s, c;
  switch (fn) {
    case PT_FN_INV_SQRT_LITERAL: return 1.0f / sqrtf(x);
    case PT_FN_INV_SQRT_FAST: return inv_sqrt_spec(x);
    case PT_FN_SQRT_LITERAL: return sqrtf(x);
    case PT_FN_SQRT_FAST: return sqrt_cr_f32(x);
    case PT_FN_SIN: pt_sincos(x, s, c); return s;
    case PT_FN_COS: pt_sincos(x, s, c); return c;
    case PT_FN_UNIFORM: return uniform_from_u32(__float_as_uint(x));
    case PT_FN_ONEMINUS_LITERAL: return (float)sqrt(1.0 - (double)(x * x));
    case PT_FN_ONEMINUS_FAST: return (float)sqrt_cr(1.0 - (double)(x * x));
    case PT_FN_ONEMINUS_F32: {
      bool bad = false;
      const float v = oneminus_f32_nb(x, bad);
      return bad ? (float)sqrt(1.0 - (double)(x * x)) : v;
    }
    case PT_FN_ONEMINUS_F32_FLAG: {
      bool bad = false;
      (void)oneminus_f32_nb(x, bad);
      return bad ? 1.0f : 0.0f;
    }
    case PT_FN_ZERO: return 0.0f;
    case PT_FN_UNIFORM_LITERAL: return uniform_from_u32_literal(__float_as_uint(x));
    case PT_FN_SIN_LITERAL: pt_sincos_literal(x, s, c); return s;
    case PT_FN_COS_LITERAL: pt_sincos_literal(x, s, c); return c;
    default: return __builtin_nanf("");
  }
}

__global__ void __launch_bounds__(256) unary_map_kernel(int fn, const float* in, float* out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = eval_unary(fn, in[i]);
}

// result[0] = number of bit patterns where fn_a != fn_b (NaN == NaN), result[1] = one such pattern
__global__ void __launch_bounds__(256) unary_compare_kernel(int fn_a, int fn_b, uint32_t first, uint64_t count,
                                                            unsigned long long* result) {
  unsigned long long bad = 0, example = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (uint64_t)gridDim.x * 256) {
    const uint32_t bits = first + (uint32_t)i;
    const float x = __uint_as_float(bits);
    const float a = eval_unary(fn_a, x), b = eval_unary(fn_b, x);
    const bool same = (__float_as_uint(a) == __float_as_uint(b)) || (a != a && b != b);
    if (!same) {
      bad++;
      example = bits;
    }
  }
  if (bad) {
    atomicAdd(&result[0], bad);
    atomicExch(&result[1], example);
  }
}

// the kernels' division by a sample count against the division: result[0] = mismatches, [1] = dividend bits, [2] = count
__global__ void __launch_bounds__(256) div_compare_kernel(uint32_t n_first, uint32_t n_count, uint32_t first, uint64_t count,
                                                          unsigned long long* result) {
  unsigned long long bad = 0, ex_bits = 0, ex_n = 0;
  for (uint32_t k = 0; k < n_count; k++) {
    const uint32_t n = n_first + k;
    const float nf = (float)n;
    const float y = (n <= (uint32_t)kRcpTab) ? 1.0f / nf : __builtin_nanf("");  // what SceneLds::rcpn holds for this count
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (uint64_t)gridDim.x * 256) {
      const uint32_t bits = first + (uint32_t)i;
      const float delta = __uint_as_float(bits);
      const float a = div_by_count(delta, nf, y), b = delta / nf;
      const bool same = (__float_as_uint(a) == __float_as_uint(b)) || (a != a && b != b);
      if (!same) {
        bad++;
        ex_bits = bits;
        ex_n = n;
      }
    }
  }
  if (bad) {
    atomicAdd(&result[0], bad);
    atomicExch(&result[1], ex_bits);
    atomicExch(&result[2], ex_n);
  }
}

}  // namespace pt

#define PT_HIPD(call)                                                                             \
  do {                                                                                            \
    hipError_t e_ = (call);                                                                       \
    if (e_ != hipSuccess) return pt_fail(PT_EHIP, "%s: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

extern "C" int pt_debug_unary_map(int fn, const float* d_in, float* d_out, size_t n) {
  if (fn < 0 || fn >= PT_FN_COUNT || (n && (!d_in || !d_out))) return pt_fail(PT_EINVAL, "pt_debug_unary_map: bad arguments");
  if (!n) return PT_OK;
  const unsigned grid = (unsigned)((n + 255) / 256 < 65536 ? (n + 255) / 256 : 65536);
  hipLaunchKernelGGL(pt::unary_map_kernel, dim3(grid), dim3(256), 0, 0, fn, d_in, d_out, n);
  PT_HIPD(hipGetLastError());
  PT_HIPD(hipDeviceSynchronize());
  return PT_OK;
}

extern "C" int pt_debug_unary_compare(int fn_a, int fn_b, uint32_t first_bits, uint64_t count, uint64_t* n_mismatch,
                                      uint32_t* example_bits) {
  if (fn_a < 0 || fn_a >= PT_FN_COUNT || fn_b < 0 || fn_b >= PT_FN_COUNT || !n_mismatch || count > (1ull << 32))
    return pt_fail(PT_EINVAL, "pt_debug_unary_compare: bad arguments");
  unsigned long long* d = nullptr;
  PT_HIPD(hipMalloc((void**)&d, 16));
  PT_HIPD(hipMemset(d, 0, 16));
  hipLaunchKernelGGL(pt::unary_compare_kernel, dim3(16384), dim3(256), 0, 0, fn_a, fn_b, first_bits, count, d);
  hipError_t e = hipGetLastError();
  unsigned long long h[2] = {0, 0};
  if (e == hipSuccess) e = hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
  (void)hipFree(d);
  if (e != hipSuccess) return pt_fail(PT_EHIP, "pt_debug_unary_compare: %s", hipGetErrorString(e));
  *n_mismatch = h[0];
  if (example_bits) *example_bits = (uint32_t)h[1];
  return PT_OK;
}

extern "C" int pt_debug_div_compare(uint32_t n_first, uint32_t n_count, uint32_t first_bits, uint64_t count, uint64_t* n_mismatch,
                                    uint32_t* example_bits, uint32_t* example_n) {
  if (!n_mismatch || count > (1ull << 32) || n_first < 1 || n_count < 1 || n_count > 4096)
    return pt_fail(PT_EINVAL, "pt_debug_div_compare: bad arguments");
  unsigned long long* d = nullptr;
  PT_HIPD(hipMalloc((void**)&d, 24));
  PT_HIPD(hipMemset(d, 0, 24));
  hipLaunchKernelGGL(pt::div_compare_kernel, dim3(16384), dim3(256), 0, 0, n_first, n_count, first_bits, count, d);
  hipError_t e = hipGetLastError();
  unsigned long long h[3] = {0, 0, 0};
  if (e == hipSuccess) e = hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
  (void)hipFree(d);
  if (e != hipSuccess) return pt_fail(PT_EHIP, "pt_debug_div_compare: %s", hipGetErrorString(e));
  *n_mismatch = h[0];
  if (example_bits) *example_bits = (uint32_t)h[1];
  if (example_n) *example_n = (uint32_t)h[2];
  return PT_OK;
}

// diagnostics: build variant 11's grid for a scene and return its 16-word header
extern "C" int pt_debug_grid_header(const pt_sphere* d_spheres, int n_spheres, uint32_t header_out[16]) {
  if (!d_spheres || !header_out || n_spheres < 1) return pt_fail(PT_EINVAL, "pt_debug_grid_header: bad arguments");
  uint32_t* d = nullptr;
  PT_HIPD(hipMalloc((void**)&d, pt_kernel_accel_bytes()));
  hipError_t e = pt_launch_build_grid(d_spheres, n_spheres, d, nullptr, true, nullptr);
  if (e == hipSuccess) e = hipMemcpy(header_out, d, 64, hipMemcpyDeviceToHost);
  (void)hipFree(d);
  PT_HIPD(e);
  return PT_OK;
}

// diagnostics: the whole accelerator image build_grid_kernel writes for a scene (header, out-of-grid list, cell starts, registrations,
// the pooled walk's cell table, emission data) and the byte offsets of its parts -- tests/test_grid_builder_gpu.py checks the
// registrations against the geometry in float64
extern "C" int pt_debug_grid_image(const pt_sphere* d_spheres, int n_spheres, const float* eye, int threads, uint32_t* image_out,
                                   size_t image_bytes, uint64_t layout_out[8]) {
  if (!d_spheres || !layout_out || n_spheres < 1) return pt_fail(PT_EINVAL, "pt_debug_grid_image: bad arguments");
  pt_kernel_grid_layout(n_spheres, threads, layout_out);
  const size_t need = (size_t)layout_out[0];
  if (!image_out) return PT_OK;  // (size query)
  if (image_bytes < need) return pt_fail(PT_EINVAL, "pt_debug_grid_image: the image needs %zu bytes", need);
  uint32_t* d = nullptr;
  PT_HIPD(hipMalloc((void**)&d, need));
  hipError_t e = hipMemset(d, 0, need);
  if (e == hipSuccess) e = pt_launch_build_grid(d_spheres, n_spheres, d, eye, true, nullptr, threads);
  if (e == hipSuccess) e = hipMemcpy(image_out, d, need, hipMemcpyDeviceToHost);
  (void)hipFree(d);
  PT_HIPD(e);
  return PT_OK;
}
