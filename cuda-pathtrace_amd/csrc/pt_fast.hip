// pt_fast.hip -- the TOLERANCED fast mode of the megakernel (pt_renderer_opts.fast_math = 1).
//
// Same algorithm as src/pathtrace.cu:150-257 and the same generator streams as the contract kernels, but NOT the
// numeric contract: this translation unit allows FMA contraction (what nvcc does to the reference by default) and
// replaces the expensive exactly-rounded pieces by what the hardware offers:
//   * intersectSphere (pathtrace.cu:72-91) in FP32 only, in the cancellation-free form: h = dot(d, off), c as the reference rounds it,
//     disc = h*h - a*c, q = h + sign(h) sqrt(disc), roots {-q/a, -c/q} (their product is c/a) -- no FP64 sqrt/divide;
//   * normalize = v * v_rsq_f32(dot(v,v)) (1 ulp) -- CUDA's own rsqrtf is an approximation as well;
//   * sinf/cosf(2 pi u) = v_sin_f32(u) / v_cos_f32(u) (the instructions take revolutions: no range reduction);
//   * pow(u, 0.5) = v_sqrt_f32, sqrt(1 - r*r) = v_sqrt_f32(1 - u) in FP32; luminance in FP32;
//   * the already unit-length shading normal is not normalised a second time and the Welford divisions share one
//     v_rcp_f32 per sample.
// It is reported BESIDE the bit-exact kernels, never instead of them.  Acceptance (tests/test_fast_mode_gpu.py):
// first-hit features at 1 spp agree with the contract build to float rounding, per-channel image means agree
// within the Monte-Carlo standard error, and the share of pixels whose colour differs by more than 1e-4 at equal
// seeds is bounded (a chaotic integrand: the reference compiled with and without contraction differs in ~1 % of
// the pixels).
#include "pt_footprint.h"

#include <type_traits>

#pragma clang fp contract(fast)

namespace pt {
namespace fast {

__device__ __forceinline__ float dot3(F3 a, F3 b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, a.z * b.z)); }
__device__ __forceinline__ F3 unit(F3 v) {
  const float k = __builtin_amdgcn_rsqf(dot3(v, v));
  return mk3(v.x * k, v.y * k, v.z * k);
}
__device__ __forceinline__ float lum(F3 c) { return fmaf(0.2126f, c.x, fmaf(0.7152f, c.y, 0.0722f * c.z)); }

struct Var {  // OnlineVarianceBuffer (pathtrace.cu:39-65), the division replaced by a multiplication with 1/n
  float n, mean, M2;
};
__device__ __forceinline__ void var_update(Var& w, float x, float n_new, float rcp_n) {
  w.n = n_new;
  const float delta = x - w.mean;
  w.mean = fmaf(delta, rcp_n, w.mean);
  w.M2 = fmaf(delta, x - w.mean, w.M2);
}
__device__ __forceinline__ float var_value(const Var& w) { return w.n < 2.0f ? 0.0f : w.M2 * __builtin_amdgcn_rcpf(w.n - 1.0f); }

// one sphere (pathtrace.cu:72-91), FP32 only: true = real roots exist; t = the root the reference returns
__device__ __forceinline__ float sphere_t(F3 o, F3 d, float a, float inv_a, const float4 g, float& disc, uint32_t absmask) {
  const F3 off = mk3(o.x - g.x, o.y - g.y, o.z - g.z);
  const float h = dot3(d, off);                                                      // b / 2
  // :76 in the reference's order: |off|^2 is rounded BEFORE r^2 is subtracted.  For the 1e5-radius wall spheres that
  // rounding (ulp 1024 at 1e10) moves the hit point by ~1e-3 units, and the image depends on it measurably: the
  // light is a 0.78-deep cap of a 600-radius sphere below the ceiling sphere, so its visible area follows the ceiling's
  // t to that precision.  Folding -r^2 into the fma chain is MORE accurate and shifts the mean radiance by 0.15 %
  // (9 standard errors at 512^2 x 1024 spp) away from the reference's arithmetic -- measured with tools/fast_bias.py on builds that
  // differed in this line only (exact sin/cos, exact normalisation and the FP64 sqrt(1 - r*r) changed nothing measurable).
  const float c = (off.x * off.x + off.y * off.y + off.z * off.z) - g.w;
  disc = fmaf(h, h, -a * c);                                                         // det / 4
  const float s = __builtin_amdgcn_sqrtf(disc);                                      // NaN when there is no real root
  const float nq = copysign_neg_b3(s, h, absmask) - h;  // -(h + copysign(s, h)): no cancellation (absmask in a VGPR: pt_device.h)
  const float t_big = nq * inv_a;                       // the root of larger magnitude
  const float t_small = c * __builtin_amdgcn_rcpf(nq);  // the other one: their product is c / a
  // The reference returns the smaller root if both are positive, else the positive one, else a non-positive number its caller
  // rejects (:82-88,99): as bit patterns that is one unsigned minimum -- positive floats order like their bits, and a set sign
  // bit (or a NaN: no real root) lies above all of them (pt_intersect.h, screen_sphere_oc).
  const uint32_t tb = __float_as_uint(t_big), ts = __float_as_uint(t_small);
  return __uint_as_float(tb < ts ? tb : ts);
}

// intersectScene (pathtrace.cu:93-107): nearest accepted t.  Up to 64 spheres are ranked as unsigned keys = the bits
// of t with the sphere index in the low bits (a negative discriminant or a negative t sets the sign bit = loses against
// every valid key): one v_min_u32 per sphere instead of compares and selects; the winner's t is then evaluated again
// at full precision.  Larger scenes use plain compares.
// mask (wave-uniform): the spheres to rank -- all of them, or for a primary ray what the wave's pixel footprints leave (pt_footprint.h)
// LAST: only hit/miss and the index are used by the caller (the last bounce of a path of known length): the key decides
template <int NS, bool LAST = false>
__device__ __forceinline__ bool nearest(const SceneLds& sc, int n, F3 o, F3 d, float& t_hit, int& idx, uint32_t mask = 0xFFFFFFFFu) {
  const float a = dot3(d, d);
  const float inv_a = __builtin_amdgcn_rcpf(a);
  if (NS > 0 || n <= 64) {
    const int nn = NS > 0 ? NS : n;
    if (nn <= 0) return false;
    const int ib = 32 - __builtin_clz((unsigned)(nn > 1 ? nn - 1 : 1));
    const uint32_t imask = (1u << ib) - 1u;
    uint32_t best = 0xFFFFFFFFu;
    auto rank = [&](const float4 g, int i) {
      float disc;
      const float t = sphere_t(o, d, a, inv_a, g, disc, sc.absmask);
      best = min(best, bitop3<0xBA>(__float_as_uint(t), imask, (uint32_t)i));  // (t & ~imask) | i; no root / negative t: sign bit or NaN bits
    };
    const uint32_t full = nn >= 32 ? 0xFFFFFFFFu : (1u << nn) - 1u;
    uint32_t m = __builtin_amdgcn_readfirstlane(mask) & full;
    if (NS > 0 && m != full) {
      while (m) {
        const int i = __builtin_ctz(m);
        m &= m - 1u;
        rank(sc.geom[i], i);
      }
    } else if constexpr (NS > 0) {
#pragma unroll
      for (int i = 0; i < NS; i++) rank(sc.geom[i], i);
    } else {
      for (int i = 0; i < n; i++) rank(sc.geom_uniform(i), i);
    }
    idx = (int)(best & imask);
    if constexpr (LAST) {
      t_hit = __uint_as_float(best & ~imask);  // the ranked t, low bits cleared (unused by the caller)
      return ((best & ~imask) != 0u) & (best < 0x49742400u);  // 0 < t < 1e6 on the key (sign bit set = no candidate)
    }
    float disc;
    const float t = sphere_t(o, d, a, inv_a, sc.geom_lane(idx), disc, sc.absmask);
    t_hit = t;
    return (best < 0x7F800000u) & (disc >= 0.0f) & (t > 0.0f) & (t < 1000000.0f);  // :94,:99
  }
  float best = 1000000.0f;  // :94
  int bi = -1;
  for (int i = 0; i < n; i++) {
    float disc;
    const float t = sphere_t(o, d, a, inv_a, sc.geom_uniform(i), disc, sc.absmask);
    if (disc >= 0.0f && t > 0.0f && t < best) {  // :77,:99 (NaN compares false)
      best = t;
      bi = i;
    }
  }
  t_hit = best;
  idx = bi < 0 ? 0 : bi;
  return bi >= 0;
}

#ifndef PT_FAST_MIN_WAVES
#define PT_FAST_MIN_WAVES 4  // 117 VGPRs; 5 or 6 waves per SIMD (96 / 80 VGPRs, spills) measured no faster: issue-bound like the exact kernels
#endif
template <int RNG, int NS, int NB>
__global__ void __launch_bounds__(PT_BLOCK_THREADS, PT_FAST_MIN_WAVES) pixel_kernel_fast(PixelKernelArgs a) {
  if constexpr (NS > 0) a.n_spheres = NS;
  if constexpr (NB > 0) a.max_bounces = NB;
  extern __shared__ float4 lds_scene[];
  SceneLds sc = stage_scene<false>(a.spheres, a.n_spheres, lds_scene, NS == 0 && a.n_spheres > PT_FAST_LDS_SPHERES, mk3(a.eye[0], a.eye[1], a.eye[2]), 0, false);

  const uint32_t tp = blockIdx.x * PT_BLOCK_THREADS + threadIdx.x;
  const bool active = tp < a.tile_pixels;
  const int row = a.row_begin + (int)(tp / (uint32_t)a.width);
  const int col = (int)(tp % (uint32_t)a.width);
  const uint32_t id = (uint32_t)row * (uint32_t)a.width + (uint32_t)col;  // :206

  Rng<RNG> rng;
  if constexpr (RNG == PT_RNG_XORWOW) {
    if (a.rng_state && active) {  // :212
      const uint32_t* s = a.rng_state + (size_t)tp * 6;
      rng.st = Xorwow{s[0], s[1], s[2], s[3], s[4], s[5]};
    } else {
      xorwow_init(rng.st, (uint64_t)id + a.seed);  // :265
    }
  } else {
    rng.k0 = (uint32_t)a.seed;
    rng.k1 = (uint32_t)(a.seed >> 32) ^ a.frame;
    rng.pix = id;
  }
  const F3 B0 = mk3(a.basis[0], a.basis[1], a.basis[2]), B1 = mk3(a.basis[3], a.basis[4], a.basis[5]);
  const F3 B2 = mk3(a.basis[6], a.basis[7], a.basis[8]), B3 = mk3(a.basis[9], a.basis[10], a.basis[11]);
  const F3 eye = mk3(a.eye[0], a.eye[1], a.eye[2]);
  const float inv_w = __builtin_amdgcn_rcpf((float)a.width), inv_h = __builtin_amdgcn_rcpf((float)a.height);

  Var var[4] = {{0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f}};
  F3 Lc = mk3(0, 0, 0), Ln = mk3(0, 0, 0), La = mk3(0, 0, 0);
  float Ld = 0.0f;

  // the spheres this wave's primary rays can return at all (pt_footprint.h; the same exclusion the exact kernels use)
  uint32_t prim_mask = 0xFFFFFFFFu;
  if constexpr (NS > 0) {
    if (a.spp >= 8) {
      auto dir_at = [&](float sx, float sy) {
        sx *= inv_h;
        sy *= inv_w;
        return lerp(lerp(B0, B1, sy), lerp(B2, B3, sy), 1.0f - sx);
      };
      const uint32_t mine = active ? primary_candidates(sc, a.n_spheres, (float)row, (float)col, dir_at) : 0u;
      uint32_t wm = 0u;
#pragma unroll
      for (int j = 0; j < NS; j++) wm |= (__builtin_amdgcn_ballot_w64(((mine >> j) & 1u) != 0u) != 0ull) ? (1u << j) : 0u;
      prim_mask = wm;
    }
  }

  const int spp = active ? a.spp : 0;
  // issue priority by progress, as in the exact kernels (pt_kernel.hip: why)
  const bool by_progress = a.spp >= PT_PRIO_MIN_SPP;
  const int q1 = a.spp / 4, q2 = a.spp / 2, q3 = a.spp - a.spp / 4;
  for (int i = 0; i < spp; i++) {  // :219
    if (by_progress) {
      const int iu = __builtin_amdgcn_readfirstlane(i);
      if ((iu & 15) == 0) {
        if (iu < q1) __builtin_amdgcn_s_setprio(3);
        else if (iu < q2) __builtin_amdgcn_s_setprio(2);
        else if (iu < q3) __builtin_amdgcn_s_setprio(1);
        else __builtin_amdgcn_s_setprio(0);
      }
    }
    rng.begin_sample((uint32_t)i);
    float sx = (float)row, sy = (float)col;  // :221-226
    if (a.spp != 1) {
      float jx, jy;
      rng.jitter(jx, jy);
      sx += jx - 0.5f;
      sy += jy - 0.5f;
    }
    sx *= inv_h;
    sy *= inv_w;
    const F3 t0 = mk3(fmaf(sy, B1.x - B0.x, B0.x), fmaf(sy, B1.y - B0.y, B0.y), fmaf(sy, B1.z - B0.z, B0.z));
    const F3 t1 = mk3(fmaf(sy, B3.x - B2.x, B2.x), fmaf(sy, B3.y - B2.y, B2.y), fmaf(sy, B3.z - B2.z, B2.z));
    const float u = 1.0f - sx;
    F3 d = mk3(fmaf(u, t1.x - t0.x, t0.x), fmaf(u, t1.y - t0.y, t0.y), fmaf(u, t1.z - t0.z, t0.z));  // :229
    F3 o = eye;
    F3 color = mk3(0.0f, 0.0f, 0.0f), mask = mk3(1.0f, 1.0f, 1.0f);
    bool escaped = false;
    auto bounce = [&](int n, auto last) -> bool {  // :155-196; false = the ray left the scene
      float t;
      int idx;
      if (!nearest<NS, decltype(last)::value>(sc, a.n_spheres, o, d, t, idx, n == 0 ? prim_mask : 0xFFFFFFFFu)) return false;
      const float4 g = sc.geom_lane(idx);
      F3 emis, scol;
      fetch_material(sc, idx, emis, scol);
      const F3 pos = mk3(fmaf(d.x, t, o.x), fmaf(d.y, t, o.y), fmaf(d.z, t, o.z));  // :163
      F3 normal = unit(mk3(pos.x - g.x, pos.y - g.y, pos.z - g.z));                 // :164
      if (!(dot3(normal, d) < 0.0f)) normal = mk3(-normal.x, -normal.y, -normal.z);  // :166
      const F3 me = mask * emis;
      if (n == 0)  // :171-174
        color = color + mk3(clampf(me.x, 0.0f, 1.0f), clampf(me.y, 0.0f, 1.0f), clampf(me.z, 0.0f, 1.0f));
      else
        color = color + me;
      mask = mask * scol;  // :175
      o = mk3(fmaf(normal.x, 0.05f, pos.x), fmaf(normal.y, 0.05f, pos.y), fmaf(normal.z, 0.05f, pos.z));  // :178
      float u_az, u_el;
      rng.bounce(n, u_az, u_el);
      // getCosineWeightedNormal (:126-136) around the unit normal
      const F3 o1 = unit(ortho_vector(normal));
      const F3 o2 = cross(normal, o1);
      const float ry = __builtin_amdgcn_sqrtf(u_el), om = __builtin_amdgcn_sqrtf(1.0f - u_el);
      const float cs = __builtin_amdgcn_cosf(u_az) * om, sn = __builtin_amdgcn_sinf(u_az) * om;  // arguments in revolutions
      d = unit(mk3(fmaf(o1.x, cs, fmaf(o2.x, sn, normal.x * ry)), fmaf(o1.y, cs, fmaf(o2.y, sn, normal.y * ry)),
                   fmaf(o1.z, cs, fmaf(o2.z, sn, normal.z * ry))));  // :180
      if (n == 0) {  // :187-195
        Ln = Ln + normal;
        La = La + scol;
        Ld += t;
        const float nn = var[1].n + 1.0f, r = __builtin_amdgcn_rcpf(nn);
        var_update(var[1], lum(normal), nn, r);
        var_update(var[2], lum(scol), nn, r);
        var_update(var[3], t, nn, r);
      }
      return true;
    };
    if constexpr (NB > 0) {
#pragma unroll
      for (int n = 0; n < NB - 1; n++) {
        if (!escaped && !bounce(n, std::false_type{})) escaped = true;
      }
      // the last bounce of a path of known length (more than one: the first hit's depth needs its t)
      if (NB > 1) { if (!escaped && !bounce(NB - 1, std::true_type{})) escaped = true; }
      else { if (!escaped && !bounce(NB - 1, std::false_type{})) escaped = true; }
    } else {
      for (int n = 0; n < a.max_bounces && !escaped; n++)
        if (!bounce(n, std::false_type{})) escaped = true;
    }
    Lc = Lc + color;  // :159 / :198
    if (!escaped) {   // :200
      const float nn = var[0].n + 1.0f;
      var_update(var[0], lum(color), nn, __builtin_amdgcn_rcpf(nn));
    }
  }
  if (by_progress) __builtin_amdgcn_s_setprio(0);

  const float rs = __builtin_amdgcn_rcpf((float)a.spp);  // :234-237
  const float px[14] = {Lc.x * rs, Lc.y * rs, Lc.z * rs, Ln.x * rs, Ln.y * rs, Ln.z * rs, La.x * rs, La.y * rs, La.z * rs, Ld * rs,
                        var_value(var[0]), var_value(var[1]), var_value(var[2]), var_value(var[3])};
  if (a.vertices && active) store_display_vertex(a.vertices + (size_t)tp * 3, a.width, row, col, px[0], px[1], px[2]);
  if (a.planar) {
    if (active) {
#pragma unroll
      for (int c = 0; c < 14; c++) a.out[(size_t)c * a.tile_pixels + tp] = px[c];
    }
  } else {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool wave_full = (__builtin_amdgcn_ballot_w64(active) == ~0ull) && ((reinterpret_cast<uintptr_t>(a.out) & 15u) == 0u);
    if (wave_full) {  // the wave's 64 pixels are one contiguous 3584-byte span: transpose through LDS, 16-byte stores
      float* wl = reinterpret_cast<float*>(lds_scene + a.scene_lds_f4) + wave * (64 * 14);
#pragma unroll
      for (int c = 0; c < 14; c++) wl[lane * 14 + c] = px[c];
      __builtin_amdgcn_wave_barrier();
      const float4* src = reinterpret_cast<const float4*>(wl);
      float4* dst = reinterpret_cast<float4*>(a.out + (size_t)(tp - (uint32_t)lane) * 14);
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int q = lane + 64 * k;
        if (q < 224) dst[q] = src[q];
      }
    } else if (active) {
      float* op = a.out + (size_t)tp * 14;
#pragma unroll
      for (int c = 0; c < 14; c++) op[c] = px[c];
    }
  }
  if constexpr (RNG == PT_RNG_XORWOW) {
    if (a.rng_state && active) {  // :256
      uint32_t* s = a.rng_state + (size_t)tp * 6;
      s[0] = rng.st.d; s[1] = rng.st.v0; s[2] = rng.st.v1; s[3] = rng.st.v2; s[4] = rng.st.v3; s[5] = rng.st.v4;
    }
  }
}

}  // namespace fast
}  // namespace pt

typedef void (*fast_kernel_fn)(PixelKernelArgs);

static fast_kernel_fn select_fast(int rng_mode, int n, int max_bounces) {
  const bool philox = rng_mode == PT_RNG_PHILOX;
  if (n == 9 && max_bounces == 5)  // the reference's configuration (Scene.h:23, pathtrace.cu:7) as compile-time constants
    return philox ? pt::fast::pixel_kernel_fast<PT_RNG_PHILOX, 9, 5> : pt::fast::pixel_kernel_fast<PT_RNG_XORWOW, 9, 5>;
  return philox ? pt::fast::pixel_kernel_fast<PT_RNG_PHILOX, 0, 0> : pt::fast::pixel_kernel_fast<PT_RNG_XORWOW, 0, 0>;
}

const void* pt_fast_kernel_symbol(int rng_mode, int n_spheres, int max_bounces) {
  return (const void*)select_fast(rng_mode, n_spheres, max_bounces);
}

size_t pt_fast_kernel_lds_bytes(int n_spheres) {
  const size_t scene = n_spheres > PT_FAST_LDS_SPHERES ? 0 : ((size_t)n_spheres * 4 + pt::kTablesF4) * sizeof(float4);
  return scene + (PT_BLOCK_THREADS / 64) * 64 * 14 * sizeof(float);
}

hipError_t pt_launch_fast_kernel(const PixelKernelArgs& a, int rng_mode, hipStream_t stream) {
  fast_kernel_fn fn = select_fast(rng_mode, a.n_spheres, a.max_bounces);
  PixelKernelArgs b = a;
  b.scene_lds_f4 = a.n_spheres > PT_FAST_LDS_SPHERES ? 0u : (uint32_t)a.n_spheres * 4u + (uint32_t)(pt::kTablesF4);
  const size_t lds = pt_fast_kernel_lds_bytes(a.n_spheres);
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, PT_LDS_BUDGET_BYTES);
    if (e != hipSuccess) return e;
  }
  const unsigned grid = (unsigned)(((uint64_t)a.tile_pixels + PT_BLOCK_THREADS - 1) / PT_BLOCK_THREADS);
  hipLaunchKernelGGL(fn, dim3(grid), dim3(PT_BLOCK_THREADS), lds, stream, b);
  return hipGetLastError();
}
