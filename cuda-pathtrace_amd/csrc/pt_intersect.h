// pt_intersect.h -- nearest-hit search over the sphere list (src/pathtrace.cu:93-107) in all its
// bit-identical forms: literal loop, screened (variants 2-6), many-sphere, and P rays at once.
#pragma once
#include "pt_scene_lds.h"

#pragma clang fp contract(off)

namespace pt {

// intersectScene: src/pathtrace.cu:93-107 -- literal loop (variants 0 and 1)
template <int VAR>
__device__ __forceinline__ bool intersect_scene_loop(const SceneLds& sc, int n, F3 o, F3 d, const RayConst& rc,
                                                     float& t_hit, int& idx) {
  float tNearest = 1000000.0f;
  float t = 0.0f;
  bool hit = false;
  // never unrolled: in the fast variants this loop is the cold fallback, inlined once per bounce, and its size
  // would otherwise push the hot path out of the instruction cache when n is a compile-time constant
#pragma clang loop unroll(disable)
  for (int i = 0; i < n; i++) {
    const float4 g = sc.geom_uniform(i);
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
    const float q = b + copysign_b3(s, b);
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

struct ScreenState {
  uint32_t k1, k2;
  bool unsure;
  uint32_t absmask;  // SceneLds::absmask
};

// Everything in the screen is kept at HALF scale -- h = dot(d, off) = b / 2, hh = h*h = bb / 4, a*c = 4ac / 4, ... --
// which spares the doubling of b per sphere: scaling by powers of two is exact, so every quantity below is exactly a
// quarter (half for the root estimates T = a*t) of the contract-scale value it stands for, and every decision
// (signs, relative margins, ranking) is the same one.
// Both return the sphere's KEY (the estimate's float bits with the index in the low bits, sign bit set = no candidate) and
// record doubts in st.unsure; screen_insert ranks a key.
__device__ __forceinline__ uint32_t screen_sphere_oc(F3 off, float c, int i, F3 d, const RayConst& rc, uint32_t imask, ScreenState& st);
__device__ __forceinline__ uint32_t screen_sphere(const float4 g, int i, F3 o, F3 d, const RayConst& rc, uint32_t imask,
                                                  ScreenState& st) {
  const F3 off = mk3(o.x - g.x, o.y - g.y, o.z - g.z);
  return screen_sphere_oc(off, dot(off, off) - g.w, i, d, rc, imask, st);
}
__device__ __forceinline__ void screen_insert(ScreenState& st, uint32_t key) {
  st.k2 = umed3(st.k1, st.k2, key);
  st.k1 = st.k1 < key ? st.k1 : key;
}
// The two smallest of the nine keys key_of(0..8) through a small network instead of nine (min, med3) insertions: three
// groups of three (min3 + med3), two merges of sorted pairs (min; min3 of the larger leader and the two runners-up):
// 12 operations instead of 18.  st.k1 / st.k2 must still be empty.
template <class F>
__device__ __forceinline__ void screen_nine(ScreenState& st, F key_of) {
  auto umin = [](uint32_t x, uint32_t y) { return x < y ? x : y; };
  auto umax = [](uint32_t x, uint32_t y) { return x > y ? x : y; };
  uint32_t lead[3], next[3];
#pragma unroll
  for (int g3 = 0; g3 < 3; g3++) {
    const uint32_t x = key_of(3 * g3), y = key_of(3 * g3 + 1), z = key_of(3 * g3 + 2);
    lead[g3] = umin(umin(x, y), z);
    next[g3] = umed3(x, y, z);
  }
  const uint32_t l = umin(lead[0], lead[1]);
  const uint32_t r2 = umin(umin(umax(lead[0], lead[1]), next[0]), next[1]);
  st.k1 = umin(l, lead[2]);
  st.k2 = umin(umin(umax(l, lead[2]), r2), next[2]);
}
// off = o - centre and c = dot(off, off) - r*r supplied by the caller (primary rays read them from SceneLds::eyeg)
__device__ __forceinline__ uint32_t screen_sphere_oc(F3 off, float c, int i, F3 d, const RayConst& rc, uint32_t imask, ScreenState& st) {
  const float h = dot(d, off);                     // b / 2
  const float hh = h * h;                          // bb / 4
  const float ac = rc.a * c;                       // 4ac / 4
  // The contract's det = RN(bb - RN(4a*c)) is used only through its sign.  dacc below rounds the exact
  // bb - 4a*c once; the two can differ in sign only when |bb - 4ac| <= 2^-24 |4ac|, and then the winner's exact
  // step decides (see the note on `unsure` below), so det itself need not be formed here.
  const float dacc = fmaf(-rc.a, c, hh);           // one rounding of the contract's exact discriminant (bb - 4ac) / 4
  const float s = __builtin_amdgcn_sqrtf(dacc);    // NaN for dacc < 0: the key below is then no candidate
  const float e = fmaf(h, h, -hh);                 // h*h - hh exactly: the rounding error of the contract's bb (/ 4)
  const float num = ac + e;
  // q = h + copysign(s, h) = |h| + s with h's sign (no cancellation); the roots (x 2a, halved: a * t) are -q and -num / q.
  // -q is formed directly (s with the sign of -h, minus h: the same float, zero signs included, since both terms share a sign),
  // so that its BITS are at hand for the unsigned minimum below without a negation of their own.
  const float TA = copysign_neg_b3(s, h, st.absmask) - h;  // root -b - sign(b) s
  const float TB = num * __builtin_amdgcn_rcpf(TA);  // root -b + sign(b) s = (4ac + (b*b - bb)) / -q
  // The reference returns the smaller root if both are positive, else the positive one, else a non-positive number that its
  // caller discards (pathtrace.cu:82-88,99).  For float bit patterns that is ONE unsigned minimum: positive floats order like
  // their bits and every negative one (sign bit set) lies above all of them -- min(bits(TA), bits(TB)) is the smaller positive
  // estimate, and a word with the sign bit set exactly when neither is positive.  (Until round 3: a median of TA, TB and an
  // infinity with the sign of -c, two instructions.)  The estimates' signs are the roots' signs unless the small root is too
  // close to 0 to classify, which is flagged below.  A negative dacc needs no bit of its own: v_sqrt_f32 of a negative number
  // is NaN and q, TA, TB with it -- bits above 0x7F800000 with either sign, which rank behind every T below the limit and fail
  // `has`.  (A negative DENORMAL dacc may be flushed to -0 by the square root: the sphere is then ranked as a tangent hit, which
  // only matters if it wins, and the winner's exact step evaluates the reference's own det -- wrongly ACCEPTING a candidate is
  // always safe here, only wrongly rejecting one is not.)
  const uint32_t ta = __float_as_uint(TA), tb = __float_as_uint(TB);
  const uint32_t tbits = ta < tb ? ta : tb;
  // key = the estimate's bits with the index in the low bits; a set sign bit = no candidate
  const uint32_t key = bitop3<0xBA>(tbits, imask, (uint32_t)i);  // (T & ~imask) | i, full rate (v_and_or_b32 is not)
  // When can the estimate not be trusted?  Only when num = ac + e has lost its leading digits (the small root then has no
  // relative accuracy, and its SIGN -- which root the reference returns -- is open) or ac is zero: since |e| <= 2^-24 hh,
  // |ac| > 2^-23 hh leaves |num| > |ac| / 2.  One fma and one compare; the floor covers hh below the normal range, where
  // the bound on e is absolute.  (NaN on either side compares false = unsure.  A non-finite ac cannot belong to a sphere the
  // reference returns, and if its garbage key wins or ties the winner's exact step sends the lane to the literal loop.)
  // Until round 2 this also required |dacc| > 2^-21 |ac| -- the band in which the reference's twice-rounded float det can
  // disagree in sign with dacc.  That is decided where it matters: a sphere wrongly taken for a hit can only do harm by
  // winning, and the winner's exact step evaluates the reference's own det; a negative dacc means a negative exact
  // discriminant, for which the reference's FP64 sqrt returns NaN and the hit is discarded (pathtrace.cu:80-88,99).
  // dacc itself is one rounding of the exact value, so the estimate is as accurate there as anywhere.
#ifndef PT_MUTANT_NO_DOUBT  // (a deliberately unsound build for tests/test_parity_gpu.py::test_ray_origins_on_sphere_surfaces to catch)
  st.unsure = st.unsure | !(fabsf(ac) > fmaf(hh, 1.1920929e-07f, 1e-30f));
#endif
  return key;
}

// PRIMARY: the ray starts at the eye the scene image was staged for -- off and c come from SceneLds::eyeg
#ifdef PT_SCREEN_STATS
__device__ unsigned long long g_screen_stats[8];
#endif
// LAST: the caller uses only the hit/miss decision and the index (the last bounce of a path of known length: emission of the
// sphere hit, nothing else -- t, the hit point and the next ray are dead).  The winner's FP64 exact step is then replaced by
// its float part and two certainty tests, see below.
template <bool NB, bool PRIMARY = false, bool LAST = false>
__device__ __forceinline__ bool intersect_scene_screened_keys(const SceneLds& sc, int n, F3 o, F3 d, const RayConst& rc,
                                                              float& t_hit, int& idx) {
  if (n <= 0) return false;
  const float Tlim = 1000000.0f * rc.a;  // the keys rank T = a*t (screen_sphere)
  const uint32_t lim_hi_bits = __float_as_uint(Tlim * 1.0000153f);
  const int ib = 32 - __builtin_clz((unsigned)(n > 1 ? n - 1 : 1));  // index bits (wave-uniform)
  const uint32_t imask = (1u << ib) - 1u;
  const float margin = 1.0f + (__builtin_ldexpf(1.0f, ib - 22) + 7.6293945e-06f);  // 2^-(22-ib) + 2^-17
  ScreenState st{0xFFFFFFFFu, 0xFFFFFFFFu, false, sc.absmask};
  auto key_of = [&](const float4 g, int i) -> uint32_t {
    if constexpr (PRIMARY) {
      const float4 e = sc.eyeg[i];
      return screen_sphere_oc(mk3(e.x, e.y, e.z), e.w, i, d, rc, imask, st);
    } else {
      return screen_sphere(g, i, o, d, rc, imask, st);
    }
  };
  auto screen = [&](const float4 g, int i) { screen_insert(st, key_of(g, i)); };
  // Manually unrolled by three (hipcc does not runtime-unroll this loop on request): the three
  // LDS reads are issued together and the three dependency chains interleave, which is what
  // keeps a lone wave busy when a small tile leaves only ~2 waves per SIMD.
  int i = 0;
  if constexpr (PRIMARY) {
    // the pixel footprints of this wave rule some spheres out for every primary ray (pt_footprint.h): rank only the rest.
    // The mask is wave-uniform, so this is a scalar loop over its set bits.
    const uint32_t full = n >= 32 ? 0xFFFFFFFFu : (1u << n) - 1u;
    uint32_t m = __builtin_amdgcn_readfirstlane(sc.prim_mask) & full;
    if (m != full) {
      while (m) {
        const int j = __builtin_ctz(m);
        m &= m - 1u;
        screen(sc.geom[j], j);
      }
      i = n;
    }
  }
#if PT_UNROLL_NINE
  if (i == 0 && n == 9) {  // the reference's scene size (Scene.h:23): constant LDS offsets and indices, no loop state
    screen_nine(st, [&](int u) { return key_of(sc.geom[u], u); });
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
  const uint32_t k1 = st.k1, k2 = st.k2;
  const bool has = k1 < lim_hi_bits;
  const float T1 = __uint_as_float(k1 & ~imask);
  bool ambiguous = st.unsure | (has & (((k2 & ~imask) <= __float_as_uint(T1 * margin)) | (T1 >= Tlim * 0.99998f)));
  bool hit = false;
  if constexpr (NB && LAST && !PRIMARY) {
    // Only "which sphere, if any" is wanted.  The exact step exists to compute t and to confirm that the winner is a hit of
    // the reference with 0 < t < 1e6; the ranking itself is the screen's.  Both confirmations can be had without FP64:
    //  * the reference takes the sphere for a hit iff its float det = RN(bb - RN(4ac)) >= 0; det and the once-rounded dacc
    //    agree in sign outside the band |dacc| <= 2^-24 |4ac| -- required here with the factor 2^-21 (the test the screen
    //    dropped per sphere, for the winner alone), and a positive dacc makes the FP64 discriminant positive too;
    //  * the estimate is within 1e-5 of the reference's t (that is what ranks the spheres), so T1 >= 1e-30 a (and
    //    T1 < 0.99998 Tlim, tested above) puts t inside (0, 1e6) with room.
    // A winner that fails either goes the usual way (exact step, literal loop).
    const int i1 = (int)(k1 & imask);
    const float4 g = sc.geom[has ? i1 : 0];
    const F3 off = mk3(o.x - g.x, o.y - g.y, o.z - g.z);
    const float h = dot(d, off);
    const float c = dot(off, off) - g.w;
    const float ac = rc.a * c;
    const float dacc = fmaf(-rc.a, c, h * h);
    const bool certain = (dacc > fabsf(ac) * 4.7683716e-07f) & (T1 >= rc.a * 1e-30f);
    ambiguous = ambiguous | (has & !certain);
    hit = has;
    t_hit = T1;  // not a distance: the caller does not use it (LAST)
    idx = i1;
    if (__builtin_expect(ambiguous, 0)) hit = intersect_scene_loop<0>(sc, n, o, d, rc, t_hit, idx);
    return hit;
  } else if constexpr (NB) {
    // straight-line: evaluate the winner unconditionally, decide afterwards
    const int i1 = (int)(k1 & imask);
    float t;
    bool bad = false;
    bool real;
    if constexpr (PRIMARY) {
      const float4 e = sc.eyeg[has ? i1 : 0];
      real = intersect_sphere_nb_oc(mk3(e.x, e.y, e.z), e.w, d, rc, t, bad);
    } else {
      real = intersect_sphere_nb(o, d, rc, sc.geom[has ? i1 : 0], t, bad);
    }
    const bool good = real & (t >= kMinGoodT) & (t < 1000000.0f);  // (quotient_to_float_nb relies on this range test)
#ifdef PT_SCREEN_STATS  // instrumentation build only (tools/screen_stats.py): how often, and why, a lane takes the literal loop
    {
      const bool tie = has & ((k2 & ~imask) <= __float_as_uint(T1 * margin));
      const bool lim = has & (T1 >= Tlim * 0.99998f);
      const uint64_t any = __builtin_amdgcn_ballot_w64(ambiguous | (has & (bad | !good)));
      if ((threadIdx.x & 63) == __builtin_ctzll(__builtin_amdgcn_ballot_w64(true))) {
        atomicAdd(&g_screen_stats[0], 1ull);
        atomicAdd(&g_screen_stats[1], any != 0 ? 1ull : 0ull);
      }
      atomicAdd(&g_screen_stats[2], (ambiguous | (has & (bad | !good))) ? 1ull : 0ull);
      atomicAdd(&g_screen_stats[3], st.unsure ? 1ull : 0ull);
      atomicAdd(&g_screen_stats[4], tie ? 1ull : 0ull);
      atomicAdd(&g_screen_stats[5], lim ? 1ull : 0ull);
      atomicAdd(&g_screen_stats[6], (has & bad) ? 1ull : 0ull);
      atomicAdd(&g_screen_stats[7], (has & !good) ? 1ull : 0ull);
    }
#endif
    ambiguous = ambiguous | (has & (bad | !good));
    hit = has & good;
    t_hit = t;
    idx = i1;
#ifdef PT_TIMING_ONLY_DETECT_NO_REDO  // never defined in a shipped build: the detection stays, the literal loop goes
    if (__builtin_expect(ambiguous, 0)) hit = false;
#elif !defined(PT_TIMING_ONLY_NO_ISECT_REDO)
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
// (Measured and dropped: also rejecting, before the square root, spheres behind the origin or provably
// farther than the lane's runner-up -- 4ac - 2^-22 bb > 2(1 + 2^-9) T2 |b| -- stayed bit-exact but cost
// more in the always-executed part than it saved: the conditional part already runs for few iterations.
// Also measured and dropped: two spheres per packed-FP32 instruction in the float part (43 instead of 80
// VALU instructions per four spheres, operands straight from SGPR pairs) -- 12 % SLOWER; with the plain
// form the loop already runs at the measured issue peak, 24.5 ns per wave and sphere for 21.6 issue units.)
__device__ __forceinline__ bool intersect_scene_screened_large(const SceneLds& sc, int n, F3 o, F3 d, const RayConst& rc,
                                                               float& t_hit, int& idx) {
  if (n <= 0) return false;  // empty scene: nothing to test and no sphere 0 to evaluate below (the array may be NULL)
  const float INF = __builtin_inff();
  const float Tlim = 1000000.0f * (2.0f * rc.a);
  const float Tlim_hi = Tlim * 1.0000153f;
  float T1 = INF, T2 = INF;
  int i1 = 0;
  bool unsure = false;
  struct Head {
    float b, a4c, bb, dacc;
  };
  auto head = [&](const float4 g) {  // the contract's float part
    const F3 off = mk3(o.x - g.x, o.y - g.y, o.z - g.z);
    Head h;
    h.b = 2.0f * dot(d, off);
    const float c = dot(off, off) - g.w;
    h.bb = h.b * h.b;
    h.a4c = rc.a4 * c;
    // One rounding of the exact bb - 4ac: has the sign of the double discriminant of pathtrace.cu:80-81.  The
    // float determinant of :77 can differ in sign only when |bb - 4ac| <= 2^-24 |4ac|; the tail flags that.
    h.dacc = fmaf(-rc.a4, c, h.bb);
    return h;
  };
  auto tail = [&](const Head& h, int i) {  // estimate + ranking, only when some lane really hits
    const bool cand = (int)__float_as_uint(h.dacc) >= 0;
    if (__builtin_amdgcn_ballot_w64(cand) == 0) return;
    const float s = __builtin_amdgcn_sqrtf(h.dacc);
    const float e = fmaf(h.b, h.b, -h.bb);
    const float num = h.a4c + e;
    const float TA = copysign_neg_b3(s, h.b, sc.absmask) - h.b;  // -q, q = b + copysign(s, b) (screen_sphere_oc)
    const float TB = num * __builtin_amdgcn_rcpf(TA);
    // the smaller positive root = the unsigned minimum of the bit patterns; sign bit set / NaN bits (dacc < 0) if there is none
    const uint32_t ta = __float_as_uint(TA), tb = __float_as_uint(TB);
    const uint32_t tbits = ta < tb ? ta : tb;
    const float T = __uint_as_float(tbits);
    const bool ok = tbits < __float_as_uint(Tlim_hi);
    unsure = unsure | (cand & !(fabsf(h.a4c) > fmaf(h.bb, 1.1920929e-07f, 1e-30f)));  // see screen_sphere_oc
    const float Te = ok ? T : INF;
    const bool c1 = Te < T1, c2 = Te < T2;
    T2 = c1 ? T1 : (c2 ? Te : T2);
    i1 = c1 ? i : i1;
    T1 = c1 ? Te : T1;
  };
  // Four spheres per iteration; one combined test skips all four conditional tails.
  int i = 0;
  auto four = [&](const float4 g0, const float4 g1, const float4 g2, const float4 g3) {
    const Head h0 = head(g0), h1 = head(g1), h2 = head(g2), h3 = head(g3);
    const uint32_t all = __float_as_uint(h0.dacc) & __float_as_uint(h1.dacc) & __float_as_uint(h2.dacc) & __float_as_uint(h3.dacc);
    if (__builtin_amdgcn_ballot_w64((int)all >= 0) != 0) {  // some lane has a non-negative discriminant for one of the four
      tail(h0, i);
      tail(h1, i + 1);
      tail(h2, i + 2);
      tail(h3, i + 3);
    }
  };
  if (sc.lean) {
    // Scalar loads, software-pipelined by hand: the requests for iteration k+1 are issued before the
    // arithmetic of iteration k and waited for after it (the compiler would sink them to their first use
    // and expose the scalar-cache latency once per iteration), hence inline assembly for both halves.
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    const pt_sphere* base = sc.global;
    auto request = [&](u4& dst, int k) {  // {radius, x, y, z} of sphere min(k, n-1)
      const uint32_t off = (uint32_t)(k < n ? k : n - 1) * (uint32_t)sizeof(pt_sphere);
      // "+v"(o.x): the float parts below read o.x, so they are ordered after the request and the
      // scheduler cannot sink it behind them (it would, to reuse the registers of the previous batch)
      asm volatile("s_load_dwordx4 %[dst], %[base], %[off]" : [dst] "=s"(dst), "+v"(o.x) : [base] "s"(base), [off] "s"(off));
    };
    auto as_geom = [](const u4 v) {
      const float r = __uint_as_float(v.x);
      return make_float4(__uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w), r * r);
    };
    if (n >= 4) {
      u4 a0, a1, a2, a3;
      request(a0, 0);
      request(a1, 1);
      request(a2, 2);
      request(a3, 3);
      asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a0), "+s"(a1), "+s"(a2), "+s"(a3));
      for (; i + 4 <= n; i += 4) {
        u4 b0, b1, b2, b3;
        request(b0, i + 4);
        request(b1, i + 5);
        request(b2, i + 6);
        request(b3, i + 7);
        four(as_geom(a0), as_geom(a1), as_geom(a2), as_geom(a3));
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(b0), "+s"(b1), "+s"(b2), "+s"(b3));
        a0 = b0;
        a1 = b1;
        a2 = b2;
        a3 = b3;
      }
    }
  } else {
    for (; i + 4 <= n; i += 4) four(sc.geom[i], sc.geom[i + 1], sc.geom[i + 2], sc.geom[i + 3]);
  }
  for (; i < n; i++) tail(head(sc.geom_uniform(i)), i);
  const bool has = T1 < INF;
  bool ambiguous = unsure | (has & ((T2 <= T1 * 1.0000038f) | (T1 >= Tlim * 0.99998f)));
  float t;
  bool bad = false;
  const bool real = intersect_sphere_nb(o, d, rc, sc.geom_lane(i1), t, bad);
  const bool good = real & (t >= kMinGoodT) & (t < 1000000.0f);  // (quotient_to_float_nb relies on this range test)
  ambiguous = ambiguous | (has & (bad | !good));
  t_hit = t;
  idx = i1;
  bool hit = has & good;
  if (__builtin_expect(ambiguous, 0)) hit = intersect_scene_loop<0>(sc, n, o, d, rc, t_hit, idx);
  return hit;
}

template <int VAR, bool PRIMARY = false, bool LAST = false>
__device__ __forceinline__ bool intersect_scene(const SceneLds& sc, int n, F3 o, F3 d, float& t_hit, int& idx) {
  // every ray but the primary one has a unit direction: its constants come from the workgroup's table (any other a: general
  // routine).  Not in the lean layouts: their kernels mix depths in a wave (path regeneration), so both routines would run.
  const RayConst rc = (VAR >= 6 && !PRIMARY && !sc.lean) ? make_ray_const_unit(d, sc.rden1) : make_ray_const(d);
  if constexpr (VAR >= 5) {
    // Screening pays when most spheres are hit by most rays (the Cornell box: a ray inside six
    // wall spheres hits all six).  In a many-sphere scene almost every test fails `det >= 0` for
    // the whole wave and the literal loop skips its FP64 part with one wave-uniform branch.
    if (!sc.lean && (sc.small_only || n <= PT_SCREEN_MAX_SPHERES))
      return intersect_scene_screened_keys<(VAR >= 6), (PRIMARY && VAR >= 6), (LAST && VAR >= 6)>(sc, n, o, d, rc, t_hit, idx);
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

// nearest hit for P rays at once; same decisions as intersect_scene_screened_keys<true>
template <int P, bool PRIMARY = false, bool LAST = false>  // LAST: see intersect_scene_screened_keys
__device__ __forceinline__ void intersect_paths(const SceneLds& sc, int n, const F3 (&o)[P], const F3 (&d)[P],
                                                bool (&hit)[P], float (&t_hit)[P], int (&idx)[P]) {
  RayConst rc[P];
  ScreenState st[P];
#pragma unroll
  for (int p = 0; p < P; p++) {
    rc[p] = (!PRIMARY && !sc.lean) ? make_ray_const_unit(d[p], sc.rden1) : make_ray_const(d[p]);
    st[p] = ScreenState{0xFFFFFFFFu, 0xFFFFFFFFu, false, sc.absmask};
    hit[p] = false;
  }
  if (n <= 0) return;
  if (sc.lean || (!sc.small_only && n > PT_SCREEN_MAX_SPHERES)) {
#pragma unroll
    for (int p = 0; p < P; p++) hit[p] = intersect_scene_screened_large(sc, n, o[p], d[p], rc[p], t_hit[p], idx[p]);
    return;
  }
  const int ib = 32 - __builtin_clz((unsigned)(n > 1 ? n - 1 : 1));
  const uint32_t imask = (1u << ib) - 1u;
  const float margin = 1.0f + (__builtin_ldexpf(1.0f, ib - 22) + 7.6293945e-06f);
  int i = 0;
  if constexpr (PRIMARY && P == 1) {  // rank only the spheres the wave's pixel footprints leave (pt_footprint.h)
    const uint32_t full = n >= 32 ? 0xFFFFFFFFu : (1u << n) - 1u;
    uint32_t m = __builtin_amdgcn_readfirstlane(sc.prim_mask) & full;
    if (n <= 16 && m != full) {
      while (m) {
        const int j = __builtin_ctz(m);
        m &= m - 1u;
        const float4 e = sc.eyeg[j];
        screen_insert(st[0], screen_sphere_oc(mk3(e.x, e.y, e.z), e.w, j, d[0], rc[0], imask, st[0]));
      }
      i = n;
    }
  }
#if PT_UNROLL_NINE
  if (i == 0 && P == 1 && n == 9) {  // the reference's scene size: fully unrolled, constant offsets
    screen_nine(st[0], [&](int u) -> uint32_t {
      if constexpr (PRIMARY) {
        const float4 e = sc.eyeg[u];
        return screen_sphere_oc(mk3(e.x, e.y, e.z), e.w, u, d[0], rc[0], imask, st[0]);
      } else {
        return screen_sphere(sc.geom[u], u, o[0], d[0], rc[0], imask, st[0]);
      }
    });
    i = 9;
  }
#endif
  for (; i + 2 <= n; i += 2) {
    const float4 g0 = sc.geom[i], g1 = sc.geom[i + 1];
#pragma unroll
    for (int p = 0; p < P; p++) {
      screen_insert(st[p], screen_sphere(g0, i, o[p], d[p], rc[p], imask, st[p]));
      screen_insert(st[p], screen_sphere(g1, i + 1, o[p], d[p], rc[p], imask, st[p]));
    }
  }
  for (; i < n; i++) {
    const float4 g = sc.geom[i];
#pragma unroll
    for (int p = 0; p < P; p++) screen_insert(st[p], screen_sphere(g, i, o[p], d[p], rc[p], imask, st[p]));
  }
  bool ambiguous[P];
#pragma unroll
  for (int p = 0; p < P; p++) {
    const float Tlim = 1000000.0f * rc[p].a;  // the keys rank T = a*t (screen_sphere)
    const bool has = st[p].k1 < __float_as_uint(Tlim * 1.0000153f);
    const float T1 = __uint_as_float(st[p].k1 & ~imask);
    ambiguous[p] = st[p].unsure | (has & (((st[p].k2 & ~imask) <= __float_as_uint(T1 * margin)) | (T1 >= Tlim * 0.99998f)));
    const int i1 = (int)(st[p].k1 & imask);
    if constexpr (LAST && !PRIMARY) {  // hit/miss and the index only: the winner's float part and two certainty tests
      const float4 g = sc.geom[has ? i1 : 0];
      const F3 off = mk3(o[p].x - g.x, o[p].y - g.y, o[p].z - g.z);
      const float h = dot(d[p], off);
      const float c = dot(off, off) - g.w;
      const float dacc = fmaf(-rc[p].a, c, h * h);
      const bool certain = (dacc > fabsf(rc[p].a * c) * 4.7683716e-07f) & (T1 >= rc[p].a * 1e-30f);
      ambiguous[p] = ambiguous[p] | (has & !certain);
      hit[p] = has;
      t_hit[p] = T1;  // not a distance: unused by the caller
      idx[p] = i1;
      continue;
    }
    float t;
    bool bad = false;
    bool real;
    if (PRIMARY && P == 1 && n <= 16) {  // the eye image was used for the screen: use it for the exact step as well
      const float4 e = sc.eyeg[has ? i1 : 0];
      real = intersect_sphere_nb_oc(mk3(e.x, e.y, e.z), e.w, d[p], rc[p], t, bad);
    } else {
      real = intersect_sphere_nb(o[p], d[p], rc[p], sc.geom[has ? i1 : 0], t, bad);
    }
    const bool good = real & (t >= kMinGoodT) & (t < 1000000.0f);  // (quotient_to_float_nb relies on this range test)
    ambiguous[p] = ambiguous[p] | (has & (bad | !good));
    hit[p] = has & good;
    t_hit[p] = t;
    idx[p] = i1;
  }
#pragma unroll
  for (int p = 0; p < P; p++)
    if (__builtin_expect(ambiguous[p], 0)) hit[p] = intersect_scene_loop<0>(sc, n, o[p], d[p], rc[p], t_hit[p], idx[p]);
}

}  // namespace pt
