// pt_footprint.h -- which spheres can a pixel's PRIMARY rays possibly return?
//
// Every primary ray of a pixel starts at the eye and points into the pixel's jitter footprint
// (sx in (row - 0.5, row + 0.5], sy likewise: src/pathtrace.cu:221-229), a patch of directions about a milliradian wide at
// 1024^2.  For most pixels a handful of the scene's spheres can be ruled out for EVERY direction of the patch: they are
// missed, or behind the eye, or farther than a wall that is certainly hit.  The bounce-0 screen (pt_intersect.h) then ranks
// only the remaining spheres -- typically 2-3 of the reference scene's 9 for a whole wave.
//
// This changes no result, for the same reason the screen itself does not: the reference returns the nearest accepted t over all
// spheres (pathtrace.cu:93-107); a sphere that is PROVABLY not that one, for any direction of the footprint and with the
// float errors of the reference's own b, c, det allowed for, can be left out of the ranking.  Spheres that are kept go through
// the unchanged machinery (keys, ambiguity margins, exact step, literal fallback -- which always loops over ALL spheres).
// Everything below is an INEQUALITY WITH SLACK, evaluated once per pixel in plain float; a NaN or an unusual geometry makes a
// comparison false, and false always means "keep the sphere".
//
// Notation: off = eye - centre, c = |off|^2 - r^2 (the reference's own floats, SceneLds::eyeg), u_c = unit direction through
// the pixel centre, rho = angular radius of the footprint around u_c, |d| in [dmin, dmax] over the footprint (d is not
// normalised, T = a t = |d| * distance).
//  (A) MISSED.  Eye robustly outside (c >= 2^-8 |off|^2).  A direction u hits the sphere iff cos^2 angle(u, off) > c / |off|^2.
//      Every u of the footprint is within rho of u_c, so |cos angle(u, off)| <= |cos angle(u_c, off)| + rho.  If
//      (|cos_c| + rho)^2 <= (c / |off|^2) (1 - 2^-12), the exact discriminant is below -2^-12 * 4ac everywhere, 500 times the
//      rounding of the reference's float det (<= 2^-21 |4ac| given c >= 2^-8 |off|^2): det < 0, the sphere is never accepted.
//  (B) BEHIND.  Eye robustly outside and u_c . off > rho |off|: the centre is behind every ray of the footprint, both roots
//      are negative (or the far one cancels to 0): never accepted (t > 0 fails).
//  (C) FARTHER THAN A WALL THAT IS ALWAYS HIT.  A sphere the eye is robustly inside of (c <= -2^-12 |off|^2) is hit by every
//      ray, at distance dist(u) = -u.off + sqrt((u.off)^2 - c), and d(dist)/d(angle) <= dist tan(phi), phi = angle between ray and
//      surface normal, cos(phi) = sqrt((u.off)^2 - c) / r.  Such a sphere is "tame" when cos(phi_c) >= 100 rho and
//      rho (1 + 2.5 dist_c / (r cos(phi_c))) <= cos(phi_c) / 2: then cos(phi) >= cos(phi_c) / 2 on the whole footprint (the
//      normal turns by at most the hit point's travel / r), tan(phi) <= 2 / cos(phi_c), and dist stays within a factor
//      exp(2 rho / cos(phi_c)) <= 1.0203 of dist_c.  W = the tame wall with the smallest upper bound U_W = dmax * 1.0203 dist_c
//      on T (and t < 0.9e6, so that it is accepted).  Sphere k is dropped if its T cannot come below 1.07 U_W anywhere:
//      tame wall: dmin * dist_c / 1.0203 >= 1.07 U_W; eye outside: any hit is at least |off| - r away,
//      dmin (|off| - r) (1 - 2^-10) >= 1.07 U_W.  The 7 % leave 2.9 % beyond the bounds for the float errors of dist_c
//      (<= 0.1 %: c's cancellation) and of the reference's own t (<= 1e-4).
// Footprints wider than 2^-8 rad (images below ~200 pixels) keep everything.
#pragma once
#include "pt_scene_lds.h"

#pragma clang fp contract(off)

namespace pt {

// The two constants of the argument above.  PT_FOOTPRINT_MUTANT (never defined in a shipped build) makes one of them
// deliberately UNSOUND -- footprint radius x 0.3, or a wall counted as farther when it is 3 % NEARER -- to show that
// tools/footprint_soak.py notices (profiles/r02/footprint_mutants.txt).
#if defined(PT_FOOTPRINT_MUTANT) && PT_FOOTPRINT_MUTANT == 1
constexpr float kFootprintRadiusFactor = 0.3f, kFootprintMargin = 1.07f;
#elif defined(PT_FOOTPRINT_MUTANT) && PT_FOOTPRINT_MUTANT == 2
constexpr float kFootprintRadiusFactor = 1.05f, kFootprintMargin = 0.93f;
#else
constexpr float kFootprintRadiusFactor = 1.05f, kFootprintMargin = 1.07f;
#endif

// dir_at(sx, sy): the primary direction for screen position (sx, sy) in PIXEL units, by the kernel's own formula
template <typename DirFn>
__device__ __forceinline__ uint32_t primary_candidates(const SceneLds& sc, int n, float row, float col, DirFn dir_at) {
  const uint32_t all = n >= 32 ? 0xFFFFFFFFu : ((1u << n) - 1u);
  if (n > 16 || n < 2) return all;
  const F3 dc = dir_at(row, col);
  const float lc = sqrtf(dot(dc, dc));
  float dev = 0.0f, dmax = lc;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const F3 dk = dir_at(row + ((k & 1) ? 0.5f : -0.5f), col + ((k & 2) ? 0.5f : -0.5f));
    const F3 e = dk - dc;
    dev = fmaxf(dev, sqrtf(dot(e, e)));
    dmax = fmaxf(dmax, sqrtf(dot(dk, dk)));
  }
  dmax *= 1.000001f;
  const float dmin = (lc - dev) * 0.999999f;
  const float rho = kFootprintRadiusFactor * dev / lc + 1e-6f;  // 1.05 dev / lc >= asin(dev / lc) for dev / lc <= 0.3
  if (!(rho <= 0.00390625f) || !(dmin > 0.0f) || !(lc > 0.0f)) return all;  // NaN, degenerate or coarse footprint: keep all
  const F3 u = dc * (1.0f / lc);

  // pass 1: the reference wall W
  float Uw = __builtin_inff();
  float dist_c[16];
  bool tame[16];
  for (int j = 0; j < n; j++) {
    const float4 e = sc.eyeg[j];
    const float rr = sc.geom[j].w, r = sqrtf(rr);
    const F3 off = mk3(e.x, e.y, e.z);
    const float o2 = dot(off, off), c = e.w, hu = dot(u, off);
    tame[j] = false;
    dist_c[j] = 0.0f;
    if (c <= -o2 * 0.000244140625f) {  // eye robustly inside
      const float sq = sqrtf(hu * hu - c);
      const float dist = hu > 0.0f ? (-c) / (sq + hu) : (sq - hu);
      const float cp = sq / r;
      const bool ok = (cp >= 100.0f * rho) && (rho * (1.0f + 2.5f * dist / (r * cp)) <= 0.5f * cp) && (dist > 0.0f);
      tame[j] = ok;
      dist_c[j] = dist;
      const float U = dmax * 1.0203f * dist;
      if (ok && (1.0203f * dist / dmin < 900000.0f) && U < Uw) Uw = U;
    }
  }
  // pass 2: drop what provably cannot be returned
  uint32_t keep = 0u;
  for (int j = 0; j < n; j++) {
    const float4 e = sc.eyeg[j];
    const float rr = sc.geom[j].w, r = sqrtf(rr);
    const F3 off = mk3(e.x, e.y, e.z);
    const float o2 = dot(off, off), c = e.w, hu = dot(u, off);
    bool drop = false;
    if (c >= o2 * 0.00390625f) {  // eye robustly outside
      const float lo = sqrtf(o2);
      const float cosc = fabsf(hu) / lo + rho;
      drop = drop | (cosc * cosc <= (c / o2) * 0.999755859375f);                     // (A) missed
      drop = drop | (hu > rho * lo * 1.000001f);                                     // (B) behind
      drop = drop | (dmin * (lo - r) * 0.9990234375f >= kFootprintMargin * Uw);      // (C) farther than W
    } else if (tame[j]) {
      drop = drop | (dmin * dist_c[j] * (1.0f / 1.0203f) >= kFootprintMargin * Uw);  // (C)
    }
    keep |= drop ? 0u : (1u << j);
  }
  return keep;
}

}  // namespace pt
