// pt_trace.h -- one path (src/pathtrace.cu:150-201): accumulating form (trace_ray) and the
// result-returning lockstep form (trace_paths + accumulate_path) used by variants 7 and 8.
#pragma once
#include "pt_grid.h"
#include "pt_footprint.h"
#include "pt_primlist.h"

#pragma clang fp contract(off)

namespace pt {

// One iteration of the bounce loop (src/pathtrace.cu:155-196) at depth n; false = the ray left the scene
// (:157-161, the path's colour has been added to L.color and the path is over).
// PRIMARY (only ever with n == 0): o is the eye the scene image was staged for, see SceneLds::eyeg
// (bounce_once = the nearest-hit search + bounce_shade; variant 12 runs the two at different times)
// the part of the iteration after intersectScene (:156): hit/t/idx are its results
// `dead_end` (variant 13's regeneration loop; never with n == 0): the path ends after this iteration, so the next ray is not
// formed -- the generator still makes its two draws (:131), which is all of :176-180 that outlives the iteration
template <int RNG, int VAR>
__device__ __forceinline__ bool bounce_shade(TraceOutput& L, const SceneLds& sc, F3& o, F3& d, F3& color, F3& mask,
                                             Rng<RNG>& rng, Welford (&var)[4], int n, bool hit, float t, int idx, bool dead_end = false) {
  if (!hit) {  // :157-161
    L.color = L.color + color;
    return false;
  }
  // (variant 13 also has every sphere's {centre, r * r} in its LDS image; reading the winner's from there instead of the lean
  // layout's global gather was measured 0.4 % SLOWER, and the material gather as a whole costs 0.5 %: profiles/r05/cfg4_ab.txt)
  const float4 g = sc.geom_lane(idx);
  F3 emis, scol;
  float lum_col = 0.0f;
  fetch_material(sc, idx, emis, scol, n == 0 ? &lum_col : nullptr);
  F3 normal = mk3(0.0f, 0.0f, 0.0f);
  float u_az, u_el;
  if constexpr (VAR >= 6) {
    // whole geometric step speculatively with the cheap sequences, literal redo if any of them
    // met an input outside its verified domain (never observed in the Cornell box)
    rng.bounce(n, u_az, u_el);
    if (!dead_end) {
      bool bad = false;
      BounceGeom bg = bounce_geometry<true, true>(o, d, t, mk3(g.x, g.y, g.z), u_az, u_el, bad, sc.inv1, sc.absmask);
#ifndef PT_TIMING_ONLY_NO_SHADE_REDO
      if (__builtin_expect(bad, 0)) bg = bounce_geometry<false>(o, d, t, mk3(g.x, g.y, g.z), u_az, u_el, bad);
#endif
      normal = bg.normal;
      o = bg.o;
      d = bg.d;
    }
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
    if (VAR >= 6 && (!sc.lean || (VAR == 13 && PT_V13_WELFORD_TABLE))) {  // (the lean brute-force kernels keep the division; the grid kernel shares ONE count and the table)
      welford_update3(var[1], var[2], var[3], luminance(normal), lum_col, t, sc.rcpn);
    } else {
      welford_update(var[1], luminance(normal));
      welford_update(var[2], lum_col);
      welford_update(var[3], t);
    }
  }
  return true;
}

// LAST (never with n == 0): the path ends after this iteration whatever happens, so nothing but colour is produced -- o, d and
// the t handed to bounce_shade are dead, and the nearest-hit search may return any t (intersect_scene_screened_keys)
// `live` (variant 13's regeneration loop only): false for a lane whose pixel is finished -- it goes through the nearest-hit search
// as a helper of the wave's pooled tests and changes nothing of its own
template <int RNG, int VAR, bool PRIMARY = false, bool LAST = false>
__device__ __forceinline__ bool bounce_once(TraceOutput& L, const SceneLds& sc, int nsph, F3& o, F3& d, F3& color, F3& mask,
                                            Rng<RNG>& rng, Welford (&var)[4], int n, bool live = true, bool prim = false,
                                            bool dead_end = false) {
  float t = 0.0f;
  int idx = 0;
  bool hit;
  if constexpr (VAR == 11)
    hit = intersect_scene_v11(sc, nsph, o, d, t, idx);
  else if constexpr (VAR == 13)
    hit = intersect_scene_v13(sc, nsph, o, d, t, idx, live, prim, dead_end);  // prim: a primary ray of a pixel with a list (pt_primlist.h)
  else
    hit = intersect_scene<VAR, PRIMARY, LAST>(sc, nsph, o, d, t, idx);
  if (VAR == 13 && !live) return true;
  return bounce_shade<RNG, VAR>(L, sc, o, d, color, mask, rng, var, n, hit, t, idx, dead_end);
}

// trace_ray: src/pathtrace.cu:150-201
// UNROLL_MB: a bounce count known at compile time (the kernel builds for one scene size and bounce cap, pt_kernel.hip: the
// reference's MAX_BOUNCES 5, and the 8 of the interactive configuration) for which the path is emitted straight-line; the
// generic builds (0) unroll the reference's five only.
template <int RNG, int VAR, int UNROLL_MB = 0>
__device__ __forceinline__ void trace_ray(TraceOutput& L, const SceneLds& sc, int nsph, F3 o, F3 d, Rng<RNG>& rng,
                                          Welford (&var)[4], int max_bounces) {
  F3 color = mk3(0.0f, 0.0f, 0.0f);
  F3 mask = mk3(1.0f, 1.0f, 1.0f);
#if PT_UNROLL_BOUNCES
  constexpr int kMB = UNROLL_MB >= 2 ? UNROLL_MB : 5;
  if (VAR >= 6 && max_bounces == kMB) {  // straight-line, no loop state, n folds to constants
    if (!bounce_once<RNG, VAR, true>(L, sc, nsph, o, d, color, mask, rng, var, 0)) return;  // trace_ray starts at the eye
#pragma unroll
    for (int n = 1; n < kMB - 1; n++)
      if (!bounce_once<RNG, VAR>(L, sc, nsph, o, d, color, mask, rng, var, n)) return;
    if (!bounce_once<RNG, VAR, false, true>(L, sc, nsph, o, d, color, mask, rng, var, kMB - 1)) return;  // the last: colour only
  } else
#endif
  {
    for (int n = 0; n < max_bounces; n++)
      if (!bounce_once<RNG, VAR>(L, sc, nsph, o, d, color, mask, rng, var, n)) return;
  }
  L.color = L.color + color;                    // :198
  if (VAR >= 6 && !sc.lean) welford_update(var[0], luminance(color), sc.rcpn); else welford_update(var[0], luminance(color));  // :200
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
  float lum_albedo0;  // luminance(albedo0) (:193), from the scene image where that is staged per sphere (same operands, same bits)
  bool hit0;     // the primary ray hit something: first-bounce features exist (:187-195)
  bool escaped;  // the path left the scene: no colour-variance update (:157-161)
};

// trace_ray (src/pathtrace.cu:150-201) for P paths in lockstep; results are returned, not accumulated
// primary_at_zero: the paths start at the eye the scene image was staged for (true for every caller in pt_kernel.hip)
template <int RNG, int P, int UNROLL_MB = 0>  // UNROLL_MB: see trace_ray
__device__ __forceinline__ void trace_paths(PathResult (&res)[P], const SceneLds& sc, int nsph, F3 (&o)[P], F3 (&d)[P],
                                            Rng<RNG> (&rng)[P], int max_bounces, bool primary_at_zero = true) {
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
    res[p].lum_albedo0 = 0.0f;
  }
  // last: the final iteration of a path of known length -- only colour comes out of it (bounce_once's LAST)
  auto bounce = [&](int n, bool last) -> bool {  // false = every path of this lane has left the scene
    bool any = false;
#pragma unroll
    for (int p = 0; p < P; p++) any = any | alive[p];
    if (!any) return false;
    bool hit[P];
    float t[P];
    int idx[P];
    if (P == 1 && primary_at_zero && n == 0)
      intersect_paths<P, true>(sc, nsph, o, d, hit, t, idx);
    else if (P == 1 && last && n > 0)
      intersect_paths<P, false, true>(sc, nsph, o, d, hit, t, idx);
    else
      intersect_paths<P>(sc, nsph, o, d, hit, t, idx);
    // stage 1 (straight-line for all paths, so their chains interleave): materials, draws, fast geometry
    F3 centre[P], emis[P], scol[P];
    float lum_col[P];
    float u_az[P], u_el[P];
    BounceGeom bg[P];
    bool bad[P];
#pragma unroll
    for (int p = 0; p < P; p++) {
      const bool was_alive = alive[p];
      res[p].escaped = res[p].escaped | (was_alive & !hit[p]);  // :157-161
      alive[p] = was_alive & hit[p];
      const int ix = alive[p] ? idx[p] : 0;
      const float4 g = sc.geom_lane(ix);
      centre[p] = mk3(g.x, g.y, g.z);
      lum_col[p] = 0.0f;
      fetch_material(sc, ix, emis[p], scol[p], n == 0 ? &lum_col[p] : nullptr);
      u_az[p] = 0.5f;
      u_el[p] = 0.5f;
      if (alive[p]) rng[p].bounce(n, u_az[p], u_el[p]);  // a dead path draws nothing
    }
#pragma unroll
    for (int p = 0; p < P; p++) {
      bad[p] = false;
      bg[p] = bounce_geometry<true, true>(o[p], d[p], t[p], centre[p], u_az[p], u_el[p], bad[p], sc.inv1, sc.absmask);
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
          res[p].lum_albedo0 = lum_col[p];
          res[p].t0 = t[p];
        }
      }
    }
    return true;
  };
#if PT_UNROLL_BOUNCES
  constexpr int kMB = UNROLL_MB >= 2 ? UNROLL_MB : 5;
  if (P == 1 && max_bounces == kMB) {
#pragma unroll
    for (int n = 0; n < kMB; n++)
      if (!bounce(n, n == kMB - 1)) break;
  } else
#endif
  {
    for (int n = 0; n < max_bounces; n++)
      if (!bounce(n, false)) break;
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

}  // namespace pt
