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
// Two layouts (chosen by the launcher; a compile-time property of the kernel build):
//  * LDS image (scenes up to PT_SCREEN_MAX_SPHERES): geometry + materials (+ the paired image variant 3 reads);
//  * lean (many-sphere scenes): NO LDS image.  A wave64 LDS read of one 16-byte sphere is a 1 KiB
//    broadcast, and with 1000 tests per bounce that broadcast traffic -- not the arithmetic -- bounded the
//    loop (25.6 ns per wave and sphere at 16 waves per CU).  The sphere index is wave-uniform, so the
//    lean build fetches {radius, centre} with SCALAR loads straight from the caller's 40-byte structs
//    (constant address space -> s_load_dwordx4 through the scalar cache; the operands then sit in SGPRs
//    and cost no VGPR, no LDS cycle and no staging pass), and gathers the winner's geometry and material
//    per lane from global memory once per bounce.  No LDS footprint means the register file alone sets
//    the occupancy, and there is no scene-size limit.
struct GridLds;  // pt_grid.h

struct SceneLds {
  float4* geom;  // {cx, cy, cz, r*r}
  float4* mat0;  // {ex, ey, ez, colx}
  float4* mat1;  // {coly, colz, luminance(col), 0}: the albedo luminance of pathtrace.cu:193 is a per-sphere constant
  float4* eyeg;  // {eye - centre, |eye - centre|^2 - r*r}: the off and c of pathtrace.cu:73,76 for a ray that starts at the eye.
                 // Every PRIMARY ray of the frame does, so these nine subtractions and dot products per sphere are done once
                 // per workgroup instead of once per sample (same operands, same operations, same bits).
  float* inv1;   // 1.0f / sqrtf(x) for the kUnitTabSize floats around 1.0f (normalize_unit_nb, pt_device.h); every exact kernel has it
  double* rden1; // make_ray_const(d).rden for dot(d, d) = the same kUnitTabSize floats (make_ray_const_unit)
  float* rcpn;   // 1.0f / (float)k for k = 1 .. min(spp, kRcpTab) at [k - 1], NaN at [kRcpTab] (welford_update, pt_device.h)
  float4* pair;  // spheres 2p,2p+1 side by side for packed FP32: {cx0,cx1,cy0,cy1}, {cz0,cz1,rr0,rr1}  (variant 3)
  const pt_sphere* global;  // the caller's array (lean build)
  bool lean;     // compile-time constant after inlining
  bool small_only;  // compile-time constant: this kernel build is only launched for scenes up to PT_SCREEN_MAX_SPHERES
  const GridLds* grid;  // variants 11, 12, 13 only
  void* pool;           // variant 13: the workgroup's pool area (pt_grid.h: per wave a test ring and the owners' result slots)
  uint32_t prim_mask;   // wave-uniform: the spheres the bounce-0 screen of this wave has to rank (pt_footprint.h); all ones = every sphere
  uint32_t absmask;     // 0x7FFFFFFF in a VGPR (vgpr_const, pt_device.h): operand of the hot paths' v_bitop3_b32 sign transfers

  // geometry of sphere i, i wave-uniform
  __device__ __forceinline__ float4 geom_uniform(int i) const {
    if (lean) {
      typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
      typedef const __attribute__((address_space(4))) f4u* cptr;
      typedef const __attribute__((address_space(4))) char* cbytes;
      const f4u v = *(cptr)((cbytes)global + (size_t)i * sizeof(pt_sphere));  // {radius, x, y, z}: Scene.h:8-9
      return make_float4(v.y, v.z, v.w, v.x * v.x);
    }
    return geom[i];
  }
  // geometry of sphere i, i per lane
  __device__ __forceinline__ float4 geom_lane(int i) const {
    if (lean) {
      const pt_sphere* sp = global + i;
      return make_float4(sp->pos[0], sp->pos[1], sp->pos[2], sp->radius * sp->radius);
    }
    return geom[i];
  }
};

typedef float v2f __attribute__((ext_vector_type(2)));

// lean: nothing of the scene is staged (many-sphere layouts read the caller's array); the two small tables are, at the
// start of the block, unless the caller has no use for them (lean_tables = false: the fast kernel)
// float4 slots of the three small tables (inv1, rden1, rcpn) that every layout keeps
constexpr int kTablesF4 = kUnitTabSize / 4 + kUnitTabSize / 2 + (kRcpTab + 4) / 4;

template <bool WITH_PAIR>
__device__ __forceinline__ SceneLds stage_scene(const pt_sphere* __restrict__ spheres, int n, float4* lds, bool lean, F3 eye,
                                                int spp, bool lean_tables = true) {
  float4* tab = lean ? lds : lds + 4 * n;
  const bool tables = !lean | lean_tables;
  SceneLds s{lds, lds + n, lds + 2 * n, lds + 3 * n, tables ? reinterpret_cast<float*>(tab) : nullptr,
             tables ? reinterpret_cast<double*>(tab + kUnitTabSize / 4) : nullptr,
             tables ? reinterpret_cast<float*>(tab + kUnitTabSize / 4 + kUnitTabSize / 2) : nullptr, lds + 4 * n + kTablesF4,
             spheres, lean, false, nullptr, nullptr, 0xFFFFFFFFu, vgpr_const<0x7FFFFFFFu>()};
  const float qnan = __builtin_nanf("");
  if (tables) {
    for (int i = threadIdx.x; i < kUnitTabSize; i += blockDim.x)  // the literal expression of helper_math's normalize (contract C2)
      s.inv1[i] = 1.0f / sqrtf(__uint_as_float(0x3F800000u - (uint32_t)kUnitTabHalf + (uint32_t)i));
    for (int i = threadIdx.x; i < kUnitTabSize; i += blockDim.x) {  // make_ray_const's own refinement, evaluated on a = dot(d, d) itself
      const float a = __uint_as_float(0x3F800000u - (uint32_t)kUnitTabHalf + (uint32_t)i);
      s.rden1[i] = ray_const_rden(2.0 * (double)a);
    }
    const int counts = spp < kRcpTab ? spp : kRcpTab;  // a pixel's accumulators never count beyond spp (lean layouts: the grid kernel reads it)
    for (int i = threadIdx.x; i < counts; i += blockDim.x) s.rcpn[i] = 1.0f / (float)(i + 1);  // the division of :52 itself
    if (threadIdx.x == 0) s.rcpn[kRcpTab] = qnan;
  }
  if (lean) {  // nothing else is staged
    __syncthreads();
    return s;
  }
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const pt_sphere sp = spheres[i];
    const float rr = sp.radius * sp.radius;
    s.geom[i] = make_float4(sp.pos[0], sp.pos[1], sp.pos[2], rr);
    s.mat0[i] = make_float4(sp.emission[0], sp.emission[1], sp.emission[2], sp.color[0]);
    s.mat1[i] = make_float4(sp.color[1], sp.color[2], luminance(mk3(sp.color[0], sp.color[1], sp.color[2])), 0.0f);
    const F3 off = mk3(eye.x - sp.pos[0], eye.y - sp.pos[1], eye.z - sp.pos[2]);  // :73 with origin = eye
    s.eyeg[i] = make_float4(off.x, off.y, off.z, dot(off, off) - rr);             // :76
    if constexpr (WITH_PAIR) {
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
  }
  __syncthreads();
  return s;
}

// emission and colour of sphere idx (Scene.h:10-11) from whichever copy the layout keeps
__device__ __forceinline__ void fetch_material(const SceneLds& sc, int idx, F3& emis, F3& scol, float* lum_col = nullptr) {
  if (sc.lean) {
#ifdef PT_TIMING_ONLY_NO_MATERIAL  // never defined in a shipped build: what the per-bounce gather from global memory costs
    emis = mk3(0.0f, 0.0f, 0.0f);
    scol = mk3(0.5f, 0.5f, 0.5f);
    if (lum_col) *lum_col = 0.5f;
    return;
#endif
    const pt_sphere* sp = sc.global + idx;
    emis = mk3(sp->emission[0], sp->emission[1], sp->emission[2]);
    scol = mk3(sp->color[0], sp->color[1], sp->color[2]);
    if (lum_col) *lum_col = luminance(scol);
  } else {
    const float4 m0 = sc.mat0[idx];
    const float4 m1 = sc.mat1[idx];
    emis = mk3(m0.x, m0.y, m0.z);
    scol = mk3(m0.w, m1.x, m1.y);
    if (lum_col) *lum_col = m1.z;  // luminance(scol), evaluated once per sphere when the scene was staged: same operands, same bits
  }
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
