/*
 * ptcore_lab.h -- additional exports of libptcore_lab.so, the LAB build of the same sources
 * (csrc/Makefile, -DPT_BUILD_EXPERIMENTS=1): everything in ptcore.h, plus the experimental kernel
 * variants 1-5, 7, 11 and 12 (stepping stones, superseded kernels and measured negative results, HISTORY.md B.2) and the
 * diagnostic entry points below.  Loaded by the variant / exhaustive tests and the tools; the
 * product library libptcore.so exports none of this.  No reference counterpart.
 */
#ifndef PTCORE_LAB_H
#define PTCORE_LAB_H

#include "ptcore.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Scalar building blocks of the device code, evaluated elementwise on the GPU. */
enum {
  PT_FN_INV_SQRT_LITERAL = 0, /* 1.0f / sqrtf(x): helper_math normalize's rsqrtf (contract C2) */
  PT_FN_INV_SQRT_FAST = 1,    /* the 7-instruction sequence the kernel uses for the same value   */
  PT_FN_SQRT_LITERAL = 2,     /* sqrtf(x)                                                        */
  PT_FN_SQRT_FAST = 3,        /* 5-instruction correctly rounded sqrt                             */
  PT_FN_SIN = 4,              /* contract C4 sin on (0, 2*pi], the kernels' instruction sequence  */
  PT_FN_COS = 5,              /* contract C4 cos                                                  */
  PT_FN_UNIFORM = 6,          /* curand_uniform mapping of the argument's BIT PATTERN             */
  PT_FN_ONEMINUS_LITERAL = 7, /* (float)sqrt(1.0 - (double)(x*x)), pathtrace.cu:134               */
  PT_FN_ONEMINUS_FAST = 8,    /* same through the lean correctly rounded double sqrt              */
  PT_FN_ONEMINUS_F32 = 9,     /* the same without FP64 (pt_device.h, oneminus_f32_nb), the literal where it flags itself */
  PT_FN_ONEMINUS_F32_FLAG = 10, /* 1.0f where oneminus_f32_nb flags itself, else 0.0f                              */
  PT_FN_ZERO = 11,
  PT_FN_UNIFORM_LITERAL = 12, /* curand_uniform as multiply, then add (the kernels use the equivalent single fma) */
  PT_FN_SIN_LITERAL = 13,     /* contract C4 as the oracle writes it (rintf, int conversion, selects); PT_FN_SIN/COS are the  */
  PT_FN_COS_LITERAL = 14,     /* kernels' form of the same floats (magic-number rounding, bit selections)                     */
  PT_FN_COUNT = 15
};
int pt_debug_unary_map(int fn, const float* d_in, float* d_out, size_t n);
/* Compare fn_a and fn_b on the `count` consecutive float bit patterns starting at first_bits
 * (count <= 2^32); NaN results compare equal.  *n_mismatch = number of differing inputs. */
int pt_debug_unary_compare(int fn_a, int fn_b, uint32_t first_bits, uint64_t count, uint64_t* n_mismatch,
                           uint32_t* example_bits);
/* The Welford update's division by the sample count (pathtrace.cu:52) as the kernels evaluate it -- reciprocal from a table,
 * exact remainder, one correction (pt_device.h, div_by_count) -- against the division itself, for the counts
 * n_first .. n_first + n_count - 1 and the `count` consecutive float bit patterns of the dividend starting at first_bits.
 * *n_mismatch = number of (count, dividend) pairs whose quotients differ in any bit (NaN == NaN). */
int pt_debug_div_compare(uint32_t n_first, uint32_t n_count, uint32_t first_bits, uint64_t count, uint64_t* n_mismatch,
                         uint32_t* example_bits, uint32_t* example_n);
/* Diagnostics: builds the uniform grid of kernel variant 11 for a scene and returns its 64-byte header
 * {valid, nx, ny, nz, origin xyz, cell size, 1/cell size, slack, centre xyz, (2E)^2, n_big, n_items}. */
int pt_debug_grid_header(const pt_sphere* d_spheres, int n_spheres, uint32_t header_out[16]);
/* Diagnostics: the whole image the grid builder writes for a scene, as the many-sphere kernel with `threads`-wide workgroups (512
 * or 1024) would get it; `eye` = camera hint or NULL.  layout_out = {image bytes, byte offsets of: out-of-grid list (u16), cell
 * starts (u16, cells + 2), registrations (u16), pooled cell table (2 x u32 per entry), emission data; cells the starts have room
 * for; entries the pooled table may have}.  image_out NULL: only the layout is returned. */
int pt_debug_grid_image(const pt_sphere* d_spheres, int n_spheres, const float* eye, int threads, uint32_t* image_out,
                        size_t image_bytes, uint64_t layout_out[8]);
/* The automatic policy's cost model (csrc/pt_capi.hip, "which kernel for a small scene"): predicted kernel milliseconds of
 * variant 6, 8 or 9 on a tile of `waves_per_simd` one-lane waves per SIMD at `spp` samples and `bounces` bounces.  Host
 * arithmetic only (no device needed): tests/test_policy_model.py holds it against the measured sweeps under profiles/. */
int pt_debug_policy_ms(int rng_mode, int variant, double waves_per_simd, int spp, int bounces, double* ms);
/* the variant the library's own cost-model policy picks (6, 8 or 9) for a tile of `waves_per_simd` one-lane waves per SIMD */
int pt_debug_policy_choice(int rng_mode, double waves_per_simd, int spp, int bounces, int with9, int chunked, int* variant);

#ifdef __cplusplus
}
#endif
#endif /* PTCORE_LAB_H */
