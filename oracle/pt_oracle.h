/*
 * pt_oracle.h -- CPU ORACLE for the per-pixel path-trace megakernel.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's `cpu_baseline` leg may load it.  Nothing under cuda-pathtrace_amd/ or
 * include/ includes, links or calls it; the product path fails loudly without the HIP
 * library and has no CPU fallback.
 *
 * What it is: a plain-C restatement of the algorithm in the reference's
 * src/pathtrace.cu (pixel_kernel / trace_ray / intersectScene / intersectSphere /
 * getCosineWeightedNormal / OnlineVarianceBuffer / luminance / setup_random), each
 * function citing the reference file:line it follows.
 *
 * PARITY STATUS: **parity unpinned**.  The reference tree holds no tests, golden images
 * or fixtures for this path, and the reference itself cannot be
 * built in this image: src/pathtrace.cu needs the CUDA runtime headers, the cuRAND device
 * API (curand_kernel.h) and CUDA-samples helper_math.h, none of which exist here, and
 * writing stand-ins for them is not allowed.  The restatement is therefore pinned only by
 *   (1) values recorded in SURVEY.md section 8(c) / 8(a) (tests/golden/survey_kats.json),
 *   (2) analytic known answers derived from include/Scene.h + include/Camera.h in float64,
 *   (3) published known-answer vectors of the third-party generators (Philox4x32-10),
 *   (4) a second restatement of the whole path in numpy (tests/numpy_restatement.py, written from the
 *       reference's lines, no code shared with this file) that pto_render reproduces bit for bit.
 * Third-party arithmetic that is NOT in /root/reference and is restated here:
 *   - cuRAND XORWOW device API (CUDA 8.0; README.md:16): curand_init/curand/curand_uniform
 *   - CUDA samples helper_math.h (CUDA 8.0; Makefile:3): float3 ops, dot, cross, normalize,
 *     lerp, clamp
 *   - CUDA device libm sinf/cosf/powf/rsqrtf: NOT reproducible off NVIDIA hardware; this
 *     oracle DEFINES deterministic replacements (see "numeric contract" in pt_oracle.c).
 */
#ifndef PT_ORACLE_H
#define PT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Same 40-byte layout as the reference's `struct Sphere` (include/Scene.h:7-14). */
typedef struct {
  float radius;
  float pos[3];
  float emission[3];
  float color[3];
} pto_sphere;

enum { PTO_RNG_XORWOW = 0, PTO_RNG_PHILOX = 1 };

typedef struct {
  int32_t width;        /* columns (reference: output.width == output.height)            */
  int32_t height;       /* rows                                                           */
  int32_t row_begin;    /* first row rendered (tile), 0 for a full frame                  */
  int32_t row_end;      /* one past the last row rendered                                 */
  int32_t spp;          /* samples per pixel (src/pathtrace.cu:219)                       */
  int32_t max_bounces;  /* MAX_BOUNCES, 5 in the reference (src/pathtrace.cu:7)           */
  int32_t rng_mode;     /* PTO_RNG_*                                                      */
  uint32_t frame;       /* philox only: frame counter mixed into the key                  */
  uint64_t seed;        /* xorwow: added to the per-pixel id (0 = reference); philox: key */
} pto_params;

/* Render rows [row_begin,row_end).  `out` points at the first float of row `row_begin`
 * (layout [row][col][14], src/pathtrace.cu:240-254).  `rng_state` is NULL (fresh
 * generator per pixel = first Render() after the Renderer constructor) or an array of
 * 6 uint32 per tile pixel {d, v0..v4} that is read and written back
 * (src/pathtrace.cu:212,256) -- only meaningful for xorwow.
 * `n_threads` rows are distributed dynamically over that many pthreads.
 * Returns 0, or -1 on bad arguments. */
int pto_render(const pto_params* p, const pto_sphere* spheres, int n_spheres,
               const float basis[12], const float eye[3], float* out,
               uint32_t* rng_state, int n_threads);

/* setup_random (src/pathtrace.cu:259-266): state[6*i..] for tile pixels. */
void pto_setup_random(const pto_params* p, uint32_t* rng_state);

/* The 9 spheres of include/Scene.h:26-34. */
void pto_scene_cornell(pto_sphere out[9]);

/* Camera::updateCameraVectors + getEyeRayBasis (include/Camera.h:125-149,153-164). */
void pto_camera_basis(const float pos[3], float yaw_deg, float pitch_deg, int w, int h,
                      float basis_out[12]);
void pto_camera_basis_up(const float pos[3], float yaw_deg, float pitch_deg, const float world_up[3], int w, int h,
                         float basis_out[12]);

/* denoise_kernel (src/denoise.cu:9-29): clamp colour to [0,1], pack RGBA8 {r,g,b,1} into one float,
 * emit per pixel the vertex triple (col, width - row, packed).  in: [row][col][14], out: [row][col][3]. */
void pto_display_pack(const float* in, int width, int height, float* out);

/* Building blocks exposed for known-answer tests. */
void pto_xorwow_init(uint64_t seed, uint32_t st[6]);
uint32_t pto_xorwow_next(uint32_t st[6]);
float pto_uniform_from_u32(uint32_t x);
void pto_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
void pto_sincos(float x, float* s, float* c);
int pto_intersect_sphere(const float o[3], const float d[3], const pto_sphere* s, float* t);

#ifdef __cplusplus
}
#endif
#endif
