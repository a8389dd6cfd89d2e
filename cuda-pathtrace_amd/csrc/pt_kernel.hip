// pt_kernel.hip -- the per-pixel Monte-Carlo megakernel for gfx950 (MI355X).
//
// Replaces src/pathtrace.cu's pixel_kernel (:203-257) and setup_random (:259-266).
// One lane = one pixel, lanes of a wave64 are 64 CONSECUTIVE COLUMNS of one image row
// (the reference maps adjacent threads to adjacent rows, i.e. a 56*W-byte lane stride).
// The sphere list is staged once per workgroup into LDS; generator state, ray, throughput,
// AOV sums and the four Welford accumulators live in VGPRs for the whole frame.
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

// intersectScene: src/pathtrace.cu:93-107 -- literal loop (variants 0 and 1)
template <int VAR>
__device__ __forceinline__ bool intersect_scene_loop(const SceneLds& sc, int n, F3 o, F3 d, const RayConst& rc,
                                                     float& t_hit, int& idx) {
  float tNearest = 1000000.0f;
  float t = 0.0f;
  bool hit = false;
  for (int i = 0; i < n; i++) {
    const float4 g = sc.geom[i];
    bool h;
    if constexpr (VAR == 0)
      h = intersect_sphere(o, d, rc.a, g, t);
    else
      h = intersect_sphere_v1(o, d, rc, g, t);
    if (h && t > 0.0f && t < tNearest) {
      tNearest = t;
      hit = true;
      t_hit = t;
      idx = i;
    }
  }
  return hit;
}

// Variant 2: screen, then evaluate exactly once.
//
// Phase 1 runs the reference's float part of every sphere test (off, b, c, b*b, det: these ARE
// the contract's values and decide `det >= 0` exactly) and adds a float32 estimate T ~ 2a*t of the
// root the reference would return, from the cancellation-free forms q = b + sign(b)*s,
// roots {-q, -(4ac + (b*b - bb))/q}, with s = sqrt(fma(-4a, c, bb)) (one rounding of the
// contract's exact discriminant, which is built on the ROUNDED product bb = b*b).  For spheres that pass the flags below, |T/(2a*t_exact) - 1| < 2^-21.  The two
// smallest estimates are kept.
// Phase 2: if the runner-up is more than 2^-18 (relative) behind, the nearest sphere is decided
// and only that one runs the FP64 path (bit-identical t).  A lane is "ambiguous" -- and redoes
// the literal loop over all spheres -- when the two best are closer than that, when a root is
// too close to zero to classify its sign, when the hit is near the 1e6 acceptance
// limit (pathtrace.cu:94), or when anything is non-finite.  Ambiguous lanes are rare
// (box edges, ~1e-5 of rays) and cost only time, never a different result.
__device__ __forceinline__ bool intersect_scene_screened(const SceneLds& sc, int n, F3 o, F3 d, const RayConst& rc,
                                                         float& t_hit, int& idx) {
  const float INF = __builtin_inff();
  const float Tlim = 1000000.0f * (2.0f * rc.a);
  const float Tlim_hi = Tlim * 1.0000153f;  // 1 + 2^-16
  float T1 = INF, T2 = INF;
  int i1 = 0;
  bool unsure = false;
PT_UNROLL(PT_SCREEN_UNROLL)
  for (int i = 0; i < n; i++) {
    const float4 g = sc.geom[i];
    const F3 off = mk3(o.x - g.x, o.y - g.y, o.z - g.z);
    const float b = 2.0f * dot(d, off);
    const float c = dot(off, off) - g.w;
    const float bb = b * b;
    const float a4c = rc.a4 * c;
    const float det = bb - a4c;
    const float dacc = fmaf(-rc.a4, c, bb);
    const float s = __builtin_amdgcn_sqrtf(fmaxf(dacc, 0.0f));
    const float q = b + copysignf(s, b);
    const float TA = -q;  // = -b - sign(b)*s: the contract's own expression, no cancellation
    // the other root -b + sign(b)*s = (s*s - b*b)/q, and s*s = bb - 4ac with the ROUNDED bb of the
    // contract: s*s - b*b = -(4ac + (b*b - bb)); e = b*b - bb is exact in one fma.
    const float e = fmaf(b, b, -bb);
    const float num = a4c + e;
    const float TB = -num * __builtin_amdgcn_rcpf(q);
    const float lo = fminf(TA, TB), hi = fmaxf(TA, TB);
    const float T = lo > 0.0f ? lo : hi;
    const bool real = det >= 0.0f && dacc >= 0.0f;
    const bool ok = real && T > 0.0f && T < Tlim_hi;
    // the sign of the cancelling root is the sign of num: reliable unless num is within its own
    // rounding error (2^-24 |4ac|) of zero; NaN/inf -> unsure
    unsure = unsure || (real && !(fabsf(num) > fabsf(a4c) * 4.7683716e-07f && fabsf(T) < INF));
    const float Te = ok ? T : INF;
    const bool c1 = Te < T1, c2 = Te < T2;
    T2 = c1 ? T1 : (c2 ? Te : T2);
    i1 = c1 ? i : i1;
    T1 = c1 ? Te : T1;
  }
  bool ambiguous = unsure || (T1 < INF && (T2 <= T1 * 1.0000038f || T1 >= Tlim * 0.99998f));
  bool hit = false;
  if (!ambiguous && T1 < INF) {
    float t;
    if (intersect_sphere_v1(o, d, rc, sc.geom[i1], t) && t > 0.0f && t < 1000000.0f) {
      hit = true;
      t_hit = t;
      idx = i1;
    } else {
      ambiguous = true;  // the estimate and the exact test disagree: let the literal loop decide
    }
  }
  if (__builtin_expect(ambiguous, 0)) hit = intersect_scene_loop<1>(sc, n, o, d, rc, t_hit, idx);
  return hit;
}

// Variant 3: the same screen with the float part evaluated for TWO spheres per instruction
// (v_pk_add/mul/fma_f32).  A plain FP32 VALU op and an FP64 op both issue at 4 cycles per
// wave64 on gfx950; only packed FP32 doubles that, and the kernel is VALU-issue bound.  The
// packed operations are the contract's own mul/add sequence (no contraction), so det, b, c
// are bit-identical to the scalar path.  Flags are combined without short-circuit branches.
__device__ __forceinline__ void screen_tail(float b, float a4c, float det, float dacc, float num, float TA, float TB,
                                            float Tlim_hi, int i, float& T1, float& T2, int& i1, bool& unsure) {
  const float INF = __builtin_inff();
  const float lo = fminf(TA, TB), hi = fmaxf(TA, TB);
  const float T = lo > 0.0f ? lo : hi;
  const bool real = (det >= 0.0f) & (dacc >= 0.0f);
  const bool ok = real & (T > 0.0f) & (T < Tlim_hi);
  unsure = unsure | (real & !((fabsf(num) > fabsf(a4c) * 4.7683716e-07f) & (fabsf(T) < INF)));
  const float Te = ok ? T : INF;
  const bool c1 = Te < T1, c2 = Te < T2;
  T2 = c1 ? T1 : (c2 ? Te : T2);
  i1 = c1 ? i : i1;
  T1 = c1 ? Te : T1;
  (void)b;
}

__device__ __forceinline__ bool intersect_scene_screened_pk(const SceneLds& sc, int n, F3 o, F3 d, const RayConst& rc,
                                                            float& t_hit, int& idx) {
  const float INF = __builtin_inff();
  const float Tlim = 1000000.0f * (2.0f * rc.a);
  const float Tlim_hi = Tlim * 1.0000153f;  // 1 + 2^-16
  float T1 = INF, T2 = INF;
  int i1 = 0;
  bool unsure = false;
  const v2f ox = {o.x, o.x}, oy = {o.y, o.y}, oz = {o.z, o.z};
  const v2f dx = {d.x, d.x}, dy = {d.y, d.y}, dz = {d.z, d.z};
  const v2f a4 = {rc.a4, rc.a4};
  const int npairs = (n + 1) >> 1;
  for (int p = 0; p < npairs; p++) {
    const float4 A = sc.pair[2 * p], B = sc.pair[2 * p + 1];
    const v2f offx = ox - v2f{A.x, A.y}, offy = oy - v2f{A.z, A.w}, offz = oz - v2f{B.x, B.y};
    const v2f dd = dx * offx + dy * offy + dz * offz;
    const v2f b = dd + dd;
    const v2f c = (offx * offx + offy * offy + offz * offz) - v2f{B.z, B.w};
    const v2f bb = b * b;
    const v2f a4c = a4 * c;
    const v2f det = bb - a4c;
    const v2f dacc = __builtin_elementwise_fma(-a4, c, bb);
    const v2f e = __builtin_elementwise_fma(b, b, -bb);
    const v2f num = a4c + e;
    const v2f s = {__builtin_amdgcn_sqrtf(fmaxf(dacc.x, 0.0f)), __builtin_amdgcn_sqrtf(fmaxf(dacc.y, 0.0f))};
    const v2f q = b + v2f{copysignf(s.x, b.x), copysignf(s.y, b.y)};
    const v2f r = {__builtin_amdgcn_rcpf(q.x), __builtin_amdgcn_rcpf(q.y)};
    const v2f TA = -q;
    const v2f TB = -num * r;
    screen_tail(b.x, a4c.x, det.x, dacc.x, num.x, TA.x, TB.x, Tlim_hi, 2 * p, T1, T2, i1, unsure);
    screen_tail(b.y, a4c.y, det.y, dacc.y, num.y, TA.y, TB.y, Tlim_hi, 2 * p + 1, T1, T2, i1, unsure);
  }
  bool ambiguous = unsure | ((T1 < INF) & ((T2 <= T1 * 1.0000038f) | (T1 >= Tlim * 0.99998f)));
  bool hit = false;
  if (!ambiguous && T1 < INF) {
    float t;
    if (intersect_sphere_v1(o, d, rc, sc.geom[i1], t) && t > 0.0f && t < 1000000.0f) {
      hit = true;
      t_hit = t;
      idx = i1;
    } else {
      ambiguous = true;
    }
  }
  if (__builtin_expect(ambiguous, 0)) hit = intersect_scene_loop<1>(sc, n, o, d, rc, t_hit, idx);
  return hit;
}

// Variant 5: the screen of variant 2 as straight-line code (no branches in the loop body, so
// unrolled iterations interleave) with fewer and cheaper instructions:
//  * validity (det >= 0, disc >= 0, T > 0) is read off the sign bits: a negative det, dacc or T
//    puts the candidate's key above every valid key;
//  * candidates are ranked as unsigned keys = float bits of T with the low ceil(log2 n) bits
//    replaced by the sphere index, so best / second best are one v_min_u32 + one v_med3_u32.
//    Truncating T costs 2^-(23-bits) of precision, which the ambiguity margin absorbs.
// A T of +0, a hit beyond the 1e6 limit or any estimate/exact disagreement is caught by the
// exact test of phase 2, which sends the lane to the literal loop.
__device__ __forceinline__ uint32_t umed3(uint32_t a, uint32_t b, uint32_t c) {
  uint32_t r;  // median of three unsigned values in one instruction (no builtin for the integer form)
  asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

template <bool NB>
__device__ __forceinline__ bool intersect_scene_screened_keys(const SceneLds& sc, int n, F3 o, F3 d, const RayConst& rc,
                                                              float& t_hit, int& idx) {
  if (n <= 0) return false;
  const float Tlim = 1000000.0f * (2.0f * rc.a);
  const uint32_t lim_hi_bits = __float_as_uint(Tlim * 1.0000153f);
  const int ib = 32 - __builtin_clz((unsigned)(n > 1 ? n - 1 : 1));  // index bits (wave-uniform)
  const uint32_t imask = (1u << ib) - 1u;
  const float margin = 1.0f + (__builtin_ldexpf(1.0f, ib - 22) + 7.6293945e-06f);  // 2^-(22-ib) + 2^-17
  uint32_t k1 = 0xFFFFFFFFu, k2 = 0xFFFFFFFFu;
  bool unsure = false;
  auto screen = [&](const float4 g, int i) {
    const F3 off = mk3(o.x - g.x, o.y - g.y, o.z - g.z);
    const float b = 2.0f * dot(d, off);
    const float c = dot(off, off) - g.w;
    const float bb = b * b;
    const float a4c = rc.a4 * c;
    const float det = bb - a4c;
    const float dacc = fmaf(-rc.a4, c, bb);
    const float s = __builtin_amdgcn_sqrtf(dacc);  // NaN for dacc < 0: the sign bit of dacc rejects it below
    const float q = b + copysignf(s, b);
    const float e = fmaf(b, b, -bb);
    const float num = a4c + e;
    const float TA = -q;
    const float TB = -num * __builtin_amdgcn_rcpf(q);
    const float lo = fminf(TA, TB), hi = fmaxf(TA, TB);
    const float T = lo > 0.0f ? lo : hi;
    const uint32_t dd = __float_as_uint(det) | __float_as_uint(dacc);
    const uint32_t w = dd | __float_as_uint(T);
    uint32_t key = (w & 0x80000000u) | __float_as_uint(T);
    key = (key & ~imask) | (uint32_t)i;
    unsure = unsure | (((int)dd >= 0) & !(fabsf(num) > fabsf(a4c) * 4.7683716e-07f));
    k2 = umed3(k1, k2, key);
    k1 = k1 < key ? k1 : key;
  };
  // Manually unrolled by three (hipcc does not runtime-unroll this loop on request): the three
  // LDS reads are issued together and the three dependency chains interleave, which is what
  // keeps a lone wave busy when a small tile leaves only ~2 waves per SIMD.
  int i = 0;
#if PT_UNROLL_NINE
  if (n == 9) {  // the reference's scene size (Scene.h:23): constant LDS offsets and indices, no loop state
#pragma unroll
    for (int u = 0; u < 9; u++) screen(sc.geom[u], u);
    i = 9;
  }
#endif
  for (; i + 3 <= n; i += 3) {
    const float4 g0 = sc.geom[i], g1 = sc.geom[i + 1], g2 = sc.geom[i + 2];
    screen(g0, i);
    screen(g1, i + 1);
    screen(g2, i + 2);
  }
  for (; i < n; i++) screen(sc.geom[i], i);
  const bool has = k1 < lim_hi_bits;
  const float T1 = __uint_as_float(k1 & ~imask);
  bool ambiguous = unsure | (has & (((k2 & ~imask) <= __float_as_uint(T1 * margin)) | (T1 >= Tlim * 0.99998f)));
  bool hit = false;
  if constexpr (NB) {
    // straight-line: evaluate the winner unconditionally, decide afterwards
    const int i1 = (int)(k1 & imask);
    float t;
    bool bad = false;
    const bool real = intersect_sphere_nb(o, d, rc, sc.geom[has ? i1 : 0], t, bad);
    const bool good = real & (t > 0.0f) & (t < 1000000.0f);
    ambiguous = ambiguous | (has & (bad | !good));
    hit = has & good;
    t_hit = t;
    idx = i1;
#ifndef PT_TIMING_ONLY_NO_ISECT_REDO
    if (__builtin_expect(ambiguous, 0)) hit = intersect_scene_loop<0>(sc, n, o, d, rc, t_hit, idx);
#endif
    return hit;
  } else {
    if (!ambiguous && has) {
      const int i1 = (int)(k1 & imask);
      float t;
      if (intersect_sphere_v1(o, d, rc, sc.geom[i1], t) && t > 0.0f && t < 1000000.0f) {
        hit = true;
        t_hit = t;
        idx = i1;
      } else {
        ambiguous = true;
      }
    }
    if (__builtin_expect(ambiguous, 0)) hit = intersect_scene_loop<1>(sc, n, o, d, rc, t_hit, idx);
    return hit;
  }
}

// Many-sphere scenes (BASELINE config 4): most spheres are missed by every lane of the wave, so the
// screen first runs only the contract's float part and skips the rest of the iteration with one
// wave-uniform branch when no lane has a real intersection.  Candidates keep full float precision
// (index tracked separately), the winner alone runs the FP64 path.
__device__ __forceinline__ bool intersect_scene_screened_large(const SceneLds& sc, int n, F3 o, F3 d, const RayConst& rc,
                                                               float& t_hit, int& idx) {
  const float INF = __builtin_inff();
  const float Tlim = 1000000.0f * (2.0f * rc.a);
  const float Tlim_hi = Tlim * 1.0000153f;
  float T1 = INF, T2 = INF;
  int i1 = 0;
  bool unsure = false;
  struct Head {
    float b, a4c, bb, dacc;
    uint32_t dd;
  };
  auto head = [&](const float4 g) {  // the contract's float part: decides det >= 0 exactly
    const F3 off = mk3(o.x - g.x, o.y - g.y, o.z - g.z);
    Head h;
    h.b = 2.0f * dot(d, off);
    const float c = dot(off, off) - g.w;
    h.bb = h.b * h.b;
    h.a4c = rc.a4 * c;
    const float det = h.bb - h.a4c;
    h.dacc = fmaf(-rc.a4, c, h.bb);
    h.dd = __float_as_uint(det) | __float_as_uint(h.dacc);
    return h;
  };
  auto tail = [&](const Head& h, int i) {  // estimate + ranking, only when some lane really hits
    if (__builtin_amdgcn_ballot_w64((int)h.dd >= 0) == 0) return;
    const float s = __builtin_amdgcn_sqrtf(h.dacc);
    const float q = h.b + copysignf(s, h.b);
    const float e = fmaf(h.b, h.b, -h.bb);
    const float num = h.a4c + e;
    const float TA = -q;
    const float TB = -num * __builtin_amdgcn_rcpf(q);
    const float lo = fminf(TA, TB), hi = fmaxf(TA, TB);
    const float T = lo > 0.0f ? lo : hi;
    const bool ok = ((int)(h.dd | __float_as_uint(T)) >= 0) & (T < Tlim_hi);
    unsure = unsure | (((int)h.dd >= 0) & !(fabsf(num) > fabsf(h.a4c) * 4.7683716e-07f));
    const float Te = ok ? T : INF;
    const bool c1 = Te < T1, c2 = Te < T2;
    T2 = c1 ? T1 : (c2 ? Te : T2);
    i1 = c1 ? i : i1;
    T1 = c1 ? Te : T1;
  };
  int i = 0;
  for (; i + 4 <= n; i += 4) {  // four LDS reads and four float parts in flight, then the conditional tails
    const float4 g0 = sc.geom[i], g1 = sc.geom[i + 1], g2 = sc.geom[i + 2], g3 = sc.geom[i + 3];
    const Head h0 = head(g0), h1 = head(g1), h2 = head(g2), h3 = head(g3);
    tail(h0, i);
    tail(h1, i + 1);
    tail(h2, i + 2);
    tail(h3, i + 3);
  }
  for (; i < n; i++) tail(head(sc.geom[i]), i);
  const bool has = T1 < INF;
  bool ambiguous = unsure | (has & ((T2 <= T1 * 1.0000038f) | (T1 >= Tlim * 0.99998f)));
  float t;
  bool bad = false;
  const bool real = intersect_sphere_nb(o, d, rc, sc.geom[i1], t, bad);
  const bool good = real & (t > 0.0f) & (t < 1000000.0f);
  ambiguous = ambiguous | (has & (bad | !good));
  t_hit = t;
  idx = i1;
  bool hit = has & good;
  if (__builtin_expect(ambiguous, 0)) hit = intersect_scene_loop<0>(sc, n, o, d, rc, t_hit, idx);
  return hit;
}

template <int VAR>
__device__ __forceinline__ bool intersect_scene(const SceneLds& sc, int n, F3 o, F3 d, float& t_hit, int& idx) {
  const RayConst rc = make_ray_const(d);
  if constexpr (VAR >= 5) {
    // Screening pays when most spheres are hit by most rays (the Cornell box: a ray inside six
    // wall spheres hits all six).  In a many-sphere scene almost every test fails `det >= 0` for
    // the whole wave and the literal loop skips its FP64 part with one wave-uniform branch.
    if (n <= PT_SCREEN_MAX_SPHERES) return intersect_scene_screened_keys<(VAR >= 6)>(sc, n, o, d, rc, t_hit, idx);
    if constexpr (VAR >= 6) return intersect_scene_screened_large(sc, n, o, d, rc, t_hit, idx);
    return intersect_scene_loop<1>(sc, n, o, d, rc, t_hit, idx);
  }
  if constexpr (VAR == 3)
    return intersect_scene_screened_pk(sc, n, o, d, rc, t_hit, idx);
  else if constexpr (VAR == 2 || VAR == 4)
    return intersect_scene_screened(sc, n, o, d, rc, t_hit, idx);
  else
    return intersect_scene_loop<VAR>(sc, n, o, d, rc, t_hit, idx);
}

// trace_ray: src/pathtrace.cu:150-201
template <int RNG, int VAR>
__device__ __forceinline__ void trace_ray(TraceOutput& L, const SceneLds& sc, int nsph, F3 o, F3 d, Rng<RNG>& rng,
                                          Welford (&var)[4], int max_bounces) {
  F3 color = mk3(0.0f, 0.0f, 0.0f);
  F3 mask = mk3(1.0f, 1.0f, 1.0f);
  auto bounce = [&](int n) -> bool {  // one iteration of the loop at :155; false = the ray left the scene
    float t = 0.0f;
    int idx = 0;
    if (!intersect_scene<VAR>(sc, nsph, o, d, t, idx)) {  // :157-161
      L.color = L.color + color;
      return false;
    }
    const float4 g = sc.geom[idx];
    const float4 m0 = sc.mat0[idx];
    const float4 m1 = sc.mat1[idx];
    const F3 emis = mk3(m0.x, m0.y, m0.z);
    const F3 scol = mk3(m0.w, m1.x, m1.y);
    F3 normal;
    float u_az, u_el;
    if constexpr (VAR >= 6) {
      // whole geometric step speculatively with the cheap sequences, literal redo if any of them
      // met an input outside its verified domain (never observed in the Cornell box)
      rng.bounce(n, u_az, u_el);
      bool bad = false;
      BounceGeom bg = bounce_geometry<true>(o, d, t, mk3(g.x, g.y, g.z), u_az, u_el, bad);
#ifndef PT_TIMING_ONLY_NO_SHADE_REDO
      if (__builtin_expect(bad, 0)) bg = bounce_geometry<false>(o, d, t, mk3(g.x, g.y, g.z), u_az, u_el, bad);
#endif
      normal = bg.normal;
      o = bg.o;
      d = bg.d;
    } else {
    F3 pos = o + d * t;                                // :163
    normal = pos - mk3(g.x, g.y, g.z);                 // :164
    if constexpr (VAR >= 4) normal = normalize_fast(normal); else normal = normalize(normal);
    if (!(dot(normal, d) < 0.0f)) normal = normal * -1.0f;  // :166
    o = pos + normal * 0.05f;         // :178, PUSH_RAY_ORIGIN
    rng.bounce(n, u_az, u_el);
    if constexpr (VAR >= 4)
      d = normalize_fast(cosine_weighted_fast(normal, u_az, u_el));  // :180
    else
      d = normalize(cosine_weighted(normal, u_az, u_el));
    }
    F3 me = mask * emis;
    if (n == 0)  // :171-172
      color = color + mk3(clampf(me.x, 0.0f, 1.0f), clampf(me.y, 0.0f, 1.0f), clampf(me.z, 0.0f, 1.0f));
    else  // :174
      color = color + me;
    mask = mask * scol;               // :175
    if (n == 0) {                     // :187-195
      L.normal = L.normal + normal;
      L.albedo = L.albedo + scol;
      L.depth += t;
      welford_update(var[1], luminance(normal));
      welford_update(var[2], luminance(scol));
      welford_update(var[3], t);
    }
    return true;
  };
#if PT_UNROLL_BOUNCES
  if (VAR >= 6 && max_bounces == 5) {  // the reference's MAX_BOUNCES (:7): straight-line, no loop state, n folds to constants
#pragma unroll
    for (int n = 0; n < 5; n++)
      if (!bounce(n)) return;
  } else
#endif
  {
    for (int n = 0; n < max_bounces; n++)
      if (!bounce(n)) return;
  }
  L.color = L.color + color;                    // :198
  welford_update(var[0], luminance(color));     // :200
}

// ---- variant 7: two samples of a pixel in lockstep ---------------------------------------------
// A lane traces samples 2k and 2k+1 together, so every stage has two independent dependency
// chains to interleave: what a small row tile (multi-GPU, ~2 waves per SIMD) needs, since there
// a lone wave is latency-bound.  Sample order is part of the contract (sequential generator,
// sequential float sums and Welford updates), so:
//  * xorwow: sample B's generator is A's advanced by the 2 + 2*max_bounces draws a non-escaping
//    path consumes (speculation).  If A escapes early the speculation was wrong and B is retraced
//    alone from A's true final state -- never in a closed scene, at worst 1.5x work in an open one;
//  * philox is counter-based: no speculation;
//  * results are accumulated strictly A then B, with the reference's own expressions.
struct PathResult {
  F3 color, normal0, albedo0;
  float t0;
  bool hit0;     // the primary ray hit something: first-bounce features exist (:187-195)
  bool escaped;  // the path left the scene: no colour-variance update (:157-161)
};

struct ScreenState {
  uint32_t k1, k2;
  bool unsure;
};

__device__ __forceinline__ void screen_sphere(const float4 g, int i, F3 o, F3 d, const RayConst& rc, uint32_t imask,
                                              ScreenState& st) {
  const F3 off = mk3(o.x - g.x, o.y - g.y, o.z - g.z);
  const float b = 2.0f * dot(d, off);
  const float c = dot(off, off) - g.w;
  const float bb = b * b;
  const float a4c = rc.a4 * c;
  const float det = bb - a4c;
  const float dacc = fmaf(-rc.a4, c, bb);
  const float s = __builtin_amdgcn_sqrtf(dacc);
  const float q = b + copysignf(s, b);
  const float e = fmaf(b, b, -bb);
  const float num = a4c + e;
  const float TA = -q;
  const float TB = -num * __builtin_amdgcn_rcpf(q);
  const float lo = fminf(TA, TB), hi = fmaxf(TA, TB);
  const float T = lo > 0.0f ? lo : hi;
  const uint32_t dd = __float_as_uint(det) | __float_as_uint(dacc);
  const uint32_t w = dd | __float_as_uint(T);
  uint32_t key = (w & 0x80000000u) | __float_as_uint(T);
  key = (key & ~imask) | (uint32_t)i;
  st.unsure = st.unsure | (((int)dd >= 0) & !(fabsf(num) > fabsf(a4c) * 4.7683716e-07f));
  st.k2 = umed3(st.k1, st.k2, key);
  st.k1 = st.k1 < key ? st.k1 : key;
}

// nearest hit for P rays at once; same decisions as intersect_scene_screened_keys<true>
template <int P>
__device__ __forceinline__ void intersect_paths(const SceneLds& sc, int n, const F3 (&o)[P], const F3 (&d)[P],
                                                bool (&hit)[P], float (&t_hit)[P], int (&idx)[P]) {
  RayConst rc[P];
  ScreenState st[P];
#pragma unroll
  for (int p = 0; p < P; p++) {
    rc[p] = make_ray_const(d[p]);
    st[p] = ScreenState{0xFFFFFFFFu, 0xFFFFFFFFu, false};
    hit[p] = false;
  }
  if (n <= 0) return;
  if (n > PT_SCREEN_MAX_SPHERES) {
#pragma unroll
    for (int p = 0; p < P; p++) hit[p] = intersect_scene_loop<1>(sc, n, o[p], d[p], rc[p], t_hit[p], idx[p]);
    return;
  }
  const int ib = 32 - __builtin_clz((unsigned)(n > 1 ? n - 1 : 1));
  const uint32_t imask = (1u << ib) - 1u;
  const float margin = 1.0f + (__builtin_ldexpf(1.0f, ib - 22) + 7.6293945e-06f);
  int i = 0;
#if PT_UNROLL_NINE
  if (P == 1 && n == 9) {  // the reference's scene size: fully unrolled, constant offsets
#pragma unroll
    for (int u = 0; u < 9; u++) screen_sphere(sc.geom[u], u, o[0], d[0], rc[0], imask, st[0]);
    i = 9;
  }
#endif
  for (; i + 2 <= n; i += 2) {
    const float4 g0 = sc.geom[i], g1 = sc.geom[i + 1];
#pragma unroll
    for (int p = 0; p < P; p++) {
      screen_sphere(g0, i, o[p], d[p], rc[p], imask, st[p]);
      screen_sphere(g1, i + 1, o[p], d[p], rc[p], imask, st[p]);
    }
  }
  for (; i < n; i++) {
    const float4 g = sc.geom[i];
#pragma unroll
    for (int p = 0; p < P; p++) screen_sphere(g, i, o[p], d[p], rc[p], imask, st[p]);
  }
  bool ambiguous[P];
#pragma unroll
  for (int p = 0; p < P; p++) {
    const float Tlim = 1000000.0f * (2.0f * rc[p].a);
    const bool has = st[p].k1 < __float_as_uint(Tlim * 1.0000153f);
    const float T1 = __uint_as_float(st[p].k1 & ~imask);
    ambiguous[p] = st[p].unsure | (has & (((st[p].k2 & ~imask) <= __float_as_uint(T1 * margin)) | (T1 >= Tlim * 0.99998f)));
    const int i1 = (int)(st[p].k1 & imask);
    float t;
    bool bad = false;
    const bool real = intersect_sphere_nb(o[p], d[p], rc[p], sc.geom[has ? i1 : 0], t, bad);
    const bool good = real & (t > 0.0f) & (t < 1000000.0f);
    ambiguous[p] = ambiguous[p] | (has & (bad | !good));
    hit[p] = has & good;
    t_hit[p] = t;
    idx[p] = i1;
  }
#pragma unroll
  for (int p = 0; p < P; p++)
    if (__builtin_expect(ambiguous[p], 0)) hit[p] = intersect_scene_loop<0>(sc, n, o[p], d[p], rc[p], t_hit[p], idx[p]);
}

// trace_ray (src/pathtrace.cu:150-201) for P paths in lockstep; results are returned, not accumulated
template <int RNG, int P>
__device__ __forceinline__ void trace_paths(PathResult (&res)[P], const SceneLds& sc, int nsph, F3 (&o)[P], F3 (&d)[P],
                                            Rng<RNG> (&rng)[P], int max_bounces) {
  F3 color[P], mask[P];
  bool alive[P];
#pragma unroll
  for (int p = 0; p < P; p++) {
    color[p] = mk3(0.0f, 0.0f, 0.0f);
    mask[p] = mk3(1.0f, 1.0f, 1.0f);
    alive[p] = true;
    res[p].hit0 = false;
    res[p].escaped = false;
    res[p].normal0 = mk3(0.0f, 0.0f, 0.0f);
    res[p].albedo0 = mk3(0.0f, 0.0f, 0.0f);
    res[p].t0 = 0.0f;
  }
  auto bounce = [&](int n) -> bool {  // false = every path of this lane has left the scene
    bool any = false;
#pragma unroll
    for (int p = 0; p < P; p++) any = any | alive[p];
    if (!any) return false;
    bool hit[P];
    float t[P];
    int idx[P];
    intersect_paths<P>(sc, nsph, o, d, hit, t, idx);
    // stage 1 (straight-line for all paths, so their chains interleave): materials, draws, fast geometry
    F3 centre[P], emis[P], scol[P];
    float u_az[P], u_el[P];
    BounceGeom bg[P];
    bool bad[P];
#pragma unroll
    for (int p = 0; p < P; p++) {
      const bool was_alive = alive[p];
      res[p].escaped = res[p].escaped | (was_alive & !hit[p]);  // :157-161
      alive[p] = was_alive & hit[p];
      const int ix = alive[p] ? idx[p] : 0;
      const float4 g = sc.geom[ix];
      const float4 m0 = sc.mat0[ix];
      const float4 m1 = sc.mat1[ix];
      centre[p] = mk3(g.x, g.y, g.z);
      emis[p] = mk3(m0.x, m0.y, m0.z);
      scol[p] = mk3(m0.w, m1.x, m1.y);
      u_az[p] = 0.5f;
      u_el[p] = 0.5f;
      if (alive[p]) rng[p].bounce(n, u_az[p], u_el[p]);  // a dead path draws nothing
    }
#pragma unroll
    for (int p = 0; p < P; p++) {
      bad[p] = false;
      bg[p] = bounce_geometry<true>(o[p], d[p], t[p], centre[p], u_az[p], u_el[p], bad[p]);
    }
    // stage 2: rare literal redo, then commit
#pragma unroll
    for (int p = 0; p < P; p++)
      if (__builtin_expect(bad[p] & alive[p], 0)) bg[p] = bounce_geometry<false>(o[p], d[p], t[p], centre[p], u_az[p], u_el[p], bad[p]);
#pragma unroll
    for (int p = 0; p < P; p++) {
      const F3 me = mask[p] * emis[p];
      const F3 add = (n == 0) ? mk3(clampf(me.x, 0.0f, 1.0f), clampf(me.y, 0.0f, 1.0f), clampf(me.z, 0.0f, 1.0f)) : me;  // :171-174
      if (alive[p]) {
        color[p] = color[p] + add;
        mask[p] = mask[p] * scol[p];  // :175
        o[p] = bg[p].o;
        d[p] = bg[p].d;
        if (n == 0) {  // :187-195 (accumulated by the caller)
          res[p].hit0 = true;
          res[p].normal0 = bg[p].normal;
          res[p].albedo0 = scol[p];
          res[p].t0 = t[p];
        }
      }
    }
    return true;
  };
#if PT_UNROLL_BOUNCES
  if (P == 1 && max_bounces == 5) {
#pragma unroll
    for (int n = 0; n < 5; n++)
      if (!bounce(n)) break;
  } else
#endif
  {
    for (int n = 0; n < max_bounces; n++)
      if (!bounce(n)) break;
  }
#pragma unroll
  for (int p = 0; p < P; p++) res[p].color = color[p];
}

// what trace_ray adds to the pixel's accumulators for one finished path, in the reference's order
__device__ __forceinline__ void accumulate_path(TraceOutput& L, Welford (&var)[4], const PathResult& r) {
  if (r.hit0) {  // :187-195
    L.normal = L.normal + r.normal0;
    L.albedo = L.albedo + r.albedo0;
    L.depth += r.t0;
    welford_update(var[1], luminance(r.normal0));
    welford_update(var[2], luminance(r.albedo0));
    welford_update(var[3], r.t0);
  }
  L.color = L.color + r.color;                                   // :159 / :198
  if (!r.escaped) welford_update(var[0], luminance(r.color));    // :200
}

// pixel_kernel: src/pathtrace.cu:203-257
template <int RNG, int VAR>
__global__ void __launch_bounds__(PT_BLOCK_THREADS, PT_MIN_WAVES) PT_KERNEL_ATTR pixel_kernel(PixelKernelArgs a) {
  extern __shared__ float4 lds_scene[];
  const SceneLds sc = stage_scene(a.spheres, a.n_spheres, lds_scene);

  const uint32_t tp = blockIdx.x * PT_BLOCK_THREADS + threadIdx.x;  // pixel index inside the tile
  const bool active = tp < a.tile_pixels;  // lanes past the tile stay for the cooperative epilogue
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

  Welford var[4] = {{0, 0.0f, 0.0f}, {0, 0.0f, 0.0f}, {0, 0.0f, 0.0f}, {0, 0.0f, 0.0f}};
  TraceOutput L{mk3(0, 0, 0), mk3(0, 0, 0), mk3(0, 0, 0), 0.0f};

  auto primary_ray = [&](Rng<RNG>& g, F3& dir) {  // :221-229
    float sx = (float)row, sy = (float)col;
    if (a.spp != 1) {
      float jx, jy;
      g.jitter(jx, jy);
      sx += jx * 1.0f - 0.5f;
      sy += jy * 1.0f - 0.5f;
    }
    sx /= (float)a.height;  // contract C7
    sy /= (float)a.width;
    dir = lerp(lerp(B0, B1, sy), lerp(B2, B3, sy), 1.0f - sx);
  };

  int i = active ? 0 : a.spp;  // inactive lanes trace nothing
  if constexpr (VAR >= 7) {
    const int draws = (a.spp != 1 ? 2 : 0) + 2 * a.max_bounces;  // consumed by a path that never escapes
    for (; i + 2 <= a.spp; i += 2) {
      Rng<RNG> g[2] = {rng, rng};
      g[0].begin_sample((uint32_t)i);
      if constexpr (RNG == PT_RNG_XORWOW) xorwow_skip(g[1].st, draws);
      g[1].begin_sample((uint32_t)i + 1u);
      F3 o2[2] = {eye, eye}, d2[2];
      primary_ray(g[0], d2[0]);
      primary_ray(g[1], d2[1]);
      PathResult res[2];
      trace_paths<RNG, 2>(res, sc, a.n_spheres, o2, d2, g, a.max_bounces);
      accumulate_path(L, var, res[0]);
      if (RNG == PT_RNG_XORWOW && __builtin_expect(res[0].escaped, 0)) {
        // A consumed fewer draws than assumed: B's stream was wrong, retrace it from A's true state
        rng = g[0];
        rng.begin_sample((uint32_t)i + 1u);
        F3 dir;
        primary_ray(rng, dir);
        trace_ray<RNG, 6>(L, sc, a.n_spheres, eye, dir, rng, var, a.max_bounces);
      } else {
        accumulate_path(L, var, res[1]);
        rng = g[1];
      }
    }
  }
  for (; i < a.spp; i++) {  // :219
    rng.begin_sample((uint32_t)i);
    F3 dir;
    primary_ray(rng, dir);
    trace_ray<RNG, (VAR >= 7 ? 6 : VAR)>(L, sc, a.n_spheres, eye, dir, rng, var, a.max_bounces);  // :231
  }

  const float fs = (float)a.spp;  // :234-237
  const float px[14] = {L.color.x / fs,  L.color.y / fs,  L.color.z / fs,  L.normal.x / fs, L.normal.y / fs,
                        L.normal.z / fs, L.albedo.x / fs, L.albedo.y / fs, L.albedo.z / fs, L.depth / fs,
                        welford_variance(var[0]), welford_variance(var[1]), welford_variance(var[2]),
                        welford_variance(var[3])};  // :240-254
  // The 64 pixels of a wave are 64 consecutive columns, so their 64 x 14 floats are ONE contiguous
  // 3584-byte span of the [row][col][14] buffer: transpose through the wave's own LDS slice and write
  // it as 224 coalesced 16-byte stores (3.5 per lane) instead of 14 strided dword stores per lane.
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool wave_full = (__builtin_amdgcn_ballot_w64(active) == ~0ull) && ((reinterpret_cast<uintptr_t>(a.out) & 15u) == 0u);
  if (wave_full) {
    float* wl = reinterpret_cast<float*>(lds_scene + a.scene_lds_f4) + wave * (64 * 14);
#pragma unroll
    for (int c = 0; c < 14; c++) wl[lane * 14 + c] = px[c];
    __builtin_amdgcn_wave_barrier();  // DS operations of one wave execute in order: the reads below see these writes
    const float4* src = reinterpret_cast<const float4*>(wl);
    float4* dst = reinterpret_cast<float4*>(a.out + (size_t)(tp - (uint32_t)lane) * 14);
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int q = lane + 64 * k;
      if (q < 224) dst[q] = src[q];
    }
  } else if (active) {
    float* o = a.out + (size_t)tp * 14;
#pragma unroll
    for (int c = 0; c < 14; c++) o[c] = px[c];
  }

  if constexpr (RNG == PT_RNG_XORWOW) {
    if (a.rng_state && active) {  // :256
      uint32_t* s = a.rng_state + (size_t)tp * 6;
      s[0] = rng.st.d;
      s[1] = rng.st.v0;
      s[2] = rng.st.v1;
      s[3] = rng.st.v2;
      s[4] = rng.st.v3;
      s[5] = rng.st.v4;
    }
  }
}

// ---- variant 8: four lanes per pixel (small tiles) ----------------------------------------------
// A rank of an 8-GPU run renders 131 072 pixels = 2 waves per SIMD with one lane per pixel, which is
// latency-bound.  Here lane (pixel, s) traces samples s, s+4, s+8, ... so the same tile
// has 4x the waves.  What the contract fixes -- one sequential generator per pixel, sequential float
// sums, sequential Welford updates -- is preserved:
//  * xorwow: lane s starts from the pixel's state advanced by s*D draws and skips 3*D draws after each
//    of its samples, D = 2 + 2*max_bounces being what a non-escaping path consumes (speculation);
//  * after every round the four results are exchanged through the wave's LDS slice and applied IN
//    SAMPLE ORDER, feature-parallel: lane 0 owns the colour sums + colour variance, lane 1 normal,
//    lane 2 albedo, lane 3 depth -- the same additions and Welford updates as the one-lane kernel,
//    just done by four lanes side by side;
//  * if a sample that is not the frame's last one escapes (consumed fewer draws), every later
//    speculative state of that pixel is wrong: the records after it are discarded and the group
//    continues in sequential mode (lane 0 traces, all four still accumulate) from that sample's true
//    final state.  Never happens in a closed scene; in an open one the kernel degrades to 1/4
//    efficiency for that pixel but stays exact.
constexpr int kSplit = 4;
constexpr int kRecWords = 24;  // 4 feature blocks {v0,v1,v2,x} + flags + 6 state words, padded

template <int RNG>
__global__ void __launch_bounds__(PT_BLOCK_THREADS) pixel_kernel_split(PixelKernelArgs a) {
  extern __shared__ float4 lds_scene[];
  const SceneLds sc = stage_scene(a.spheres, a.n_spheres, lds_scene);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* xl = reinterpret_cast<float*>(lds_scene + a.scene_lds_f4) + wave * (64 * kRecWords);
  const int gbase = lane & ~(kSplit - 1);

  const uint32_t gl = blockIdx.x * PT_BLOCK_THREADS + threadIdx.x;
  const uint32_t tp = gl / kSplit;      // pixel index inside the tile
  const int s = (int)(gl % kSplit);     // sample slot AND the feature this lane accumulates
  const bool active = tp < a.tile_pixels;
  const int row = a.row_begin + (int)(tp / (uint32_t)a.width);
  const int col = (int)(tp % (uint32_t)a.width);
  const uint32_t id = (uint32_t)row * (uint32_t)a.width + (uint32_t)col;

  Rng<RNG> rng;
  const int D = (a.spp != 1 ? 2 : 0) + 2 * a.max_bounces;
  if constexpr (RNG == PT_RNG_XORWOW) {
    if (a.rng_state && active) {
      const uint32_t* p = a.rng_state + (size_t)tp * 6;
      rng.st = Xorwow{p[0], p[1], p[2], p[3], p[4], p[5]};
    } else {
      xorwow_init(rng.st, (uint64_t)id + a.seed);
    }
    xorwow_skip(rng.st, s * D);
  } else {
    rng.k0 = (uint32_t)a.seed;
    rng.k1 = (uint32_t)(a.seed >> 32) ^ a.frame;
    rng.pix = id;
  }
  Xorwow final_state = Xorwow{0, 0, 0, 0, 0, 0};
  if constexpr (RNG == PT_RNG_XORWOW) final_state = rng.st;  // spp == 0 never happens; overwritten below

  const F3 B0 = mk3(a.basis[0], a.basis[1], a.basis[2]), B1 = mk3(a.basis[3], a.basis[4], a.basis[5]);
  const F3 B2 = mk3(a.basis[6], a.basis[7], a.basis[8]), B3 = mk3(a.basis[9], a.basis[10], a.basis[11]);
  const F3 eye = mk3(a.eye[0], a.eye[1], a.eye[2]);

  float sum0 = 0.0f, sum1 = 0.0f, sum2 = 0.0f;  // this lane's feature sums (depth uses sum0 only)
  Welford w{0, 0.0f, 0.0f};
  bool seq = false;  // group-uniform: sequential mode after a failed speculation
  int base = 0;      // group-uniform: first sample of the current round

  while (base < a.spp) {
    const int k = seq ? base : base + s;
    const bool mine = active && (!seq || s == 0) && k < a.spp;
    PathResult res[1];
    res[0].color = res[0].normal0 = res[0].albedo0 = mk3(0.0f, 0.0f, 0.0f);
    res[0].t0 = 0.0f;
    res[0].hit0 = res[0].escaped = false;
    if (mine) {
      Rng<RNG> g[1] = {rng};
      g[0].begin_sample((uint32_t)k);
      float sx = (float)row, sy = (float)col;  // :221-229
      if (a.spp != 1) {
        float jx, jy;
        g[0].jitter(jx, jy);
        sx += jx * 1.0f - 0.5f;
        sy += jy * 1.0f - 0.5f;
      }
      sx /= (float)a.height;
      sy /= (float)a.width;
      F3 o1[1] = {eye}, d1[1] = {lerp(lerp(B0, B1, sy), lerp(B2, B3, sy), 1.0f - sx)};
      trace_paths<RNG, 1>(res, sc, a.n_spheres, o1, d1, g, a.max_bounces);
      rng = g[0];
    }
    // publish this lane's sample
    float* rec = xl + lane * kRecWords;
    const float lumC = luminance(res[0].color), lumN = luminance(res[0].normal0), lumA = luminance(res[0].albedo0);
    *reinterpret_cast<float4*>(rec + 0) = make_float4(res[0].color.x, res[0].color.y, res[0].color.z, lumC);
    *reinterpret_cast<float4*>(rec + 4) = make_float4(res[0].normal0.x, res[0].normal0.y, res[0].normal0.z, lumN);
    *reinterpret_cast<float4*>(rec + 8) = make_float4(res[0].albedo0.x, res[0].albedo0.y, res[0].albedo0.z, lumA);
    *reinterpret_cast<float4*>(rec + 12) = make_float4(res[0].t0, 0.0f, 0.0f, res[0].t0);
    reinterpret_cast<uint32_t*>(rec)[16] = (mine ? 1u : 0u) | (res[0].hit0 ? 2u : 0u) | (res[0].escaped ? 4u : 0u);
    if constexpr (RNG == PT_RNG_XORWOW) {
      uint32_t* ru = reinterpret_cast<uint32_t*>(rec) + 17;
      ru[0] = rng.st.d; ru[1] = rng.st.v0; ru[2] = rng.st.v1; ru[3] = rng.st.v2; ru[4] = rng.st.v3; ru[5] = rng.st.v4;
    }
    __builtin_amdgcn_wave_barrier();
    // apply the round's samples in order to this lane's feature
    bool failed = false;
    int consumed = 0;  // samples of this round that count (all of them unless a speculation failed)
#pragma unroll
    for (int j = 0; j < kSplit; j++) {
      const float* rj = xl + (gbase + j) * kRecWords;
      const uint32_t fl = reinterpret_cast<const uint32_t*>(rj)[16];
      const bool valid = ((fl & 1u) != 0u) & !failed;
      const bool hit0 = (fl & 2u) != 0u, esc = (fl & 4u) != 0u;
      const float4 v = *reinterpret_cast<const float4*>(rj + 4 * s);
      const bool en_sum = valid & ((s == 0) | hit0);                 // colour is always added (:159,:198); AOVs on a first hit (:187-190)
      const bool en_w = valid & ((s == 0) ? !esc : hit0);             // :200 / :192-194
      sum0 = en_sum ? sum0 + v.x : sum0;
      sum1 = en_sum ? sum1 + v.y : sum1;
      sum2 = en_sum ? sum2 + v.z : sum2;
      Welford wn = w;
      welford_update(wn, v.w);
      w.n = en_w ? wn.n : w.n;  // field-wise: a struct select would go through scratch
      w.mean = en_w ? wn.mean : w.mean;
      w.M2 = en_w ? wn.M2 : w.M2;
      if constexpr (RNG == PT_RNG_XORWOW) {
        if (valid) {
          const uint32_t* ru = reinterpret_cast<const uint32_t*>(rj) + 17;
          final_state = Xorwow{ru[0], ru[1], ru[2], ru[3], ru[4], ru[5]};
        }
        const int kj = seq ? base : base + j;
        if (valid & esc & !seq & (kj < a.spp - 1)) failed = true;  // later speculative states are wrong
      }
      consumed += valid ? 1 : 0;
    }
    __builtin_amdgcn_wave_barrier();  // records are rewritten next round
    if (!seq) {
      if (failed) {
        if (s == 0 && a.fail_count) atomicAdd(a.fail_count, 1u);  // lets the host stop speculating on open scenes
        seq = true;
        base += consumed;          // resume right after the escaped sample
        if constexpr (RNG == PT_RNG_XORWOW) rng.st = final_state;  // its true final state (only lane 0 traces from here on)
      } else {
        base += kSplit;
        if constexpr (RNG == PT_RNG_XORWOW) xorwow_skip(rng.st, (kSplit - 1) * D);
      }
    } else {
      base += 1;
      if constexpr (RNG == PT_RNG_XORWOW) rng.st = final_state;
    }
  }

  if (active) {  // :234-254, each lane stores its feature
    const float fs = (float)a.spp;
    float* o = a.out + (size_t)tp * 14;
    if (s < 3) {
      o[3 * s + 0] = sum0 / fs;
      o[3 * s + 1] = sum1 / fs;
      o[3 * s + 2] = sum2 / fs;
    } else {
      o[9] = sum0 / fs;
    }
    o[10 + s] = welford_variance(w);
    if constexpr (RNG == PT_RNG_XORWOW) {
      if (a.rng_state && s == 0) {  // :256
        uint32_t* p = a.rng_state + (size_t)tp * 6;
        p[0] = final_state.d; p[1] = final_state.v0; p[2] = final_state.v1;
        p[3] = final_state.v2; p[4] = final_state.v3; p[5] = final_state.v4;
      }
    }
  }
}

// setup_random: src/pathtrace.cu:259-266
__global__ void __launch_bounds__(PT_BLOCK_THREADS)
    setup_random_kernel(uint32_t* state, int width, int row_begin, uint32_t tile_pixels, uint64_t seed) {
  const uint32_t tp = blockIdx.x * PT_BLOCK_THREADS + threadIdx.x;
  if (tp >= tile_pixels) return;
  const uint32_t id = (uint32_t)(row_begin + (int)(tp / (uint32_t)width)) * (uint32_t)width + tp % (uint32_t)width;
  Xorwow s;
  xorwow_init(s, (uint64_t)id + seed);
  uint32_t* p = state + (size_t)tp * 6;
  p[0] = s.d;
  p[1] = s.v0;
  p[2] = s.v1;
  p[3] = s.v2;
  p[4] = s.v3;
  p[5] = s.v4;
}

}  // namespace pt

// ---- launchers (host) ---------------------------------------------------------------------
static inline size_t scene_lds_f4(int n) { return (size_t)n * 3 + (size_t)((n + 1) / 2) * 2; }
// scene image + one 64 x 14 float transpose slice per wave for the epilogue
static inline size_t scene_lds_bytes(int n) { return scene_lds_f4(n) * sizeof(float4) + (PT_BLOCK_THREADS / 64) * 64 * 14 * sizeof(float); }

typedef void (*pixel_kernel_fn)(PixelKernelArgs);

static inline size_t split_lds_bytes(int n) {
  return scene_lds_f4(n) * sizeof(float4) + (PT_BLOCK_THREADS / 64) * 64 * pt::kRecWords * sizeof(float);
}

static pixel_kernel_fn select_kernel(int rng_mode, int variant) {
  const bool philox = rng_mode == PT_RNG_PHILOX;
  switch (variant) {
    case 0: return philox ? pt::pixel_kernel<PT_RNG_PHILOX, 0> : pt::pixel_kernel<PT_RNG_XORWOW, 0>;
    case 1: return philox ? pt::pixel_kernel<PT_RNG_PHILOX, 1> : pt::pixel_kernel<PT_RNG_XORWOW, 1>;
    case 2: return philox ? pt::pixel_kernel<PT_RNG_PHILOX, 2> : pt::pixel_kernel<PT_RNG_XORWOW, 2>;
    case 3: return philox ? pt::pixel_kernel<PT_RNG_PHILOX, 3> : pt::pixel_kernel<PT_RNG_XORWOW, 3>;
    case 4: return philox ? pt::pixel_kernel<PT_RNG_PHILOX, 4> : pt::pixel_kernel<PT_RNG_XORWOW, 4>;
    case 5: return philox ? pt::pixel_kernel<PT_RNG_PHILOX, 5> : pt::pixel_kernel<PT_RNG_XORWOW, 5>;
    case 6: return philox ? pt::pixel_kernel<PT_RNG_PHILOX, 6> : pt::pixel_kernel<PT_RNG_XORWOW, 6>;
    case 7: return philox ? pt::pixel_kernel<PT_RNG_PHILOX, 7> : pt::pixel_kernel<PT_RNG_XORWOW, 7>;
    case 8: return philox ? pt::pixel_kernel_split<PT_RNG_PHILOX> : pt::pixel_kernel_split<PT_RNG_XORWOW>;
    default: return nullptr;
  }
}

int pt_kernel_num_variants(void) { return 9; }

const void* pt_kernel_symbol(int rng_mode, int variant) { return (const void*)select_kernel(rng_mode, variant); }

size_t pt_kernel_lds_bytes(int n_spheres, int variant) {
  return variant == 8 ? split_lds_bytes(n_spheres) : scene_lds_bytes(n_spheres);
}

int pt_kernel_max_spheres(int variant) {
  if (variant == 8)
    return (int)((PT_LDS_BUDGET_BYTES - (PT_BLOCK_THREADS / 64) * 64 * pt::kRecWords * sizeof(float)) / (4 * sizeof(float4))) - 1;
  return (int)((PT_LDS_BUDGET_BYTES - (PT_BLOCK_THREADS / 64) * 64 * 14 * sizeof(float)) / (4 * sizeof(float4))) - 1;
}

hipError_t pt_launch_pixel_kernel(const PixelKernelArgs& a, int rng_mode, int variant, hipStream_t stream) {
  if (variant == 8) {
    PixelKernelArgs b = a;
    b.scene_lds_f4 = (uint32_t)scene_lds_f4(a.n_spheres);
    const size_t lds = split_lds_bytes(a.n_spheres);
    pixel_kernel_fn fs = rng_mode == PT_RNG_PHILOX ? pt::pixel_kernel_split<PT_RNG_PHILOX> : pt::pixel_kernel_split<PT_RNG_XORWOW>;
    if (lds > 64 * 1024) {
      hipError_t e = hipFuncSetAttribute((const void*)fs, hipFuncAttributeMaxDynamicSharedMemorySize, PT_LDS_BUDGET_BYTES);
      if (e != hipSuccess) return e;
    }
    const uint64_t lanes = (uint64_t)a.tile_pixels * pt::kSplit;
    const unsigned grid = (unsigned)((lanes + PT_BLOCK_THREADS - 1) / PT_BLOCK_THREADS);
    hipLaunchKernelGGL(fs, dim3(grid), dim3(PT_BLOCK_THREADS), lds, stream, b);
    return hipGetLastError();
  }
  pixel_kernel_fn fn = select_kernel(rng_mode, variant);
  if (!fn) return hipErrorInvalidValue;
  const unsigned grid = (a.tile_pixels + PT_BLOCK_THREADS - 1) / PT_BLOCK_THREADS;
  PixelKernelArgs b = a;
  b.scene_lds_f4 = (uint32_t)scene_lds_f4(a.n_spheres);
  if (scene_lds_bytes(a.n_spheres) > 64 * 1024) {  // beyond the default dynamic-LDS limit: opt in (gfx950 has 160 KiB per CU)
    hipError_t e = hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, PT_LDS_BUDGET_BYTES);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(fn, dim3(grid), dim3(PT_BLOCK_THREADS), scene_lds_bytes(a.n_spheres), stream, b);
  return hipGetLastError();
}

hipError_t pt_launch_setup_random(uint32_t* state, int width, int row_begin, uint32_t tile_pixels, uint64_t seed,
                                  hipStream_t stream) {
  const unsigned grid = (tile_pixels + PT_BLOCK_THREADS - 1) / PT_BLOCK_THREADS;
  hipLaunchKernelGGL(pt::setup_random_kernel, dim3(grid), dim3(PT_BLOCK_THREADS), 0, stream, state, width, row_begin,
                     tile_pixels, seed);
  return hipGetLastError();
}
