// pt_scene_lds.h -- the workgroup's LDS image of the scene and the two per-pixel generators.
#pragma once
#include "pt_device.h"
#include "pt_kernel.h"

#pragma clang fp contract(off)

namespace pt {

#define PT_PRAGMA_(x) _Pragma(#x)
#define PT_UNROLL(n) PT_PRAGMA_(unroll n)

// LDS image of the scene: geometry and material split so the intersect loop touches
// 16 B per sphere with a wave-uniform address (LDS broadcast read), and the shading step
// gathers 32 B by the per-lane hit index.
struct SceneLds {
  float4* geom;  // {cx, cy, cz, r*r}
  float4* mat0;  // {ex, ey, ez, colx}
  float4* mat1;  // {coly, colz, 0, 0}
  float4* pair;  // spheres 2p,2p+1 side by side for packed FP32: {cx0,cx1,cy0,cy1}, {cz0,cz1,rr0,rr1}
};

typedef float v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ SceneLds stage_scene(const pt_sphere* __restrict__ spheres, int n, float4* lds) {
  SceneLds s{lds, lds + n, lds + 2 * n, lds + 3 * n};
  const float qnan = __builtin_nanf("");
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const pt_sphere sp = spheres[i];
    const float rr = sp.radius * sp.radius;
    s.geom[i] = make_float4(sp.pos[0], sp.pos[1], sp.pos[2], rr);
    s.mat0[i] = make_float4(sp.emission[0], sp.emission[1], sp.emission[2], sp.color[0]);
    s.mat1[i] = make_float4(sp.color[1], sp.color[2], 0.0f, 0.0f);
    float* pa = reinterpret_cast<float*>(s.pair + 2 * (i >> 1)) + (i & 1);
    pa[0] = sp.pos[0];
    pa[2] = sp.pos[1];
    pa[4] = sp.pos[2];
    pa[6] = rr;
    if ((i == n - 1) && !(i & 1)) {  // odd count: the partner slot is a NaN sphere that can never hit
      pa[1] = qnan;
      pa[3] = qnan;
      pa[5] = qnan;
      pa[7] = qnan;
    }
  }
  __syncthreads();
  return s;
}

struct TraceOutput {  // src/pathtrace.cu:24-36
  F3 color, normal, albedo;
  float depth;
};

template <int RNG>
struct Rng;

template <>
struct Rng<PT_RNG_XORWOW> {
  Xorwow st;
  __device__ __forceinline__ void begin_sample(uint32_t) {}
  __device__ __forceinline__ void jitter(float& jx, float& jy) {
    jx = uniform_from_u32(xorwow_next(st));
    jy = uniform_from_u32(xorwow_next(st));
  }
  __device__ __forceinline__ void bounce(int, float& az, float& el) {
    az = uniform_from_u32(xorwow_next(st));  // first draw -> azimuth (contract C5)
    el = uniform_from_u32(xorwow_next(st));
  }
};

template <>
struct Rng<PT_RNG_PHILOX> {
  uint32_t k0, k1, pix, sample;
  uint4 blk;
  __device__ __forceinline__ void begin_sample(uint32_t s) {
    sample = s;
    blk = philox4x32_10(make_uint4(pix, sample, 0u, 0u), k0, k1);
  }
  __device__ __forceinline__ void jitter(float& jx, float& jy) {
    jx = uniform_from_u32(blk.x);
    jy = uniform_from_u32(blk.y);
  }
  __device__ __forceinline__ void bounce(int n, float& az, float& el) {
    if (n & 1) blk = philox4x32_10(make_uint4(pix, sample, (uint32_t)((n + 1) >> 1), 0u), k0, k1);
    bool second = (n == 0) || (((n + 1) & 1) != 0);
    az = uniform_from_u32(second ? blk.z : blk.x);
    el = uniform_from_u32(second ? blk.w : blk.y);
  }
};

}  // namespace pt
