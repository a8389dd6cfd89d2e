// pt_primlist.h -- variant 13, round 5: which grid spheres can a pixel's PRIMARY rays possibly return?
//
// Every primary ray of a pixel starts at the eye and points into the pixel's jitter footprint (src/pathtrace.cu:221-229), a
// cone of directions about a milliradian wide at 1024^2.  Until round 4 each of a pixel's spp primary rays walked the grid from
// the eye again -- a fifth of the rays of the closed 1000-sphere scene, half of the open one's, where two pixels in three see
// nothing but sky and still walked the whole box 256 times.  Here the wave works out, ONCE per pixel (per sample chunk), the
// list of grid spheres the pixel's cone can touch, and writes it behind the cell table in the table's own entry format
// (pt_grid.h): bounce 0 of such a pixel then starts "in" its list, does not walk, and the pooled tests, the ranking, the exact
// step and the literal fallback run on the list's spheres exactly as they would on a cell's.  Spheres outside the grid (walls:
// GridHeader::n_big) are tested by every ray in grid_begin as before.
//
// Why no result can change: a sphere is left out of a pixel's list only if the REFERENCE ITSELF can never accept it for any
// direction d of the footprint -- its own float discriminant is negative (pathtrace.cu:77-79), or both roots are (:82-88, :99).
// What is left goes through the unchanged machinery, whose only requirement on the tested set is that it contains the sphere the
// reference returns (EXACTNESS.md A.6/A.9).  Everything below is an inequality with slack, evaluated in FP64 on the reference's own
// float operands; NaN or unusual geometry makes a comparison false, and false always means "keep the sphere" / "no list".
//
// Notation: off = eye - centre and rr = r * r as the reference rounds them (float); o2 = |off|^2, a = |d|^2, in real arithmetic
// on those floats; p_d = distance of the centre from the LINE through the eye along d:  b^2 - 4 a c = 4 a (rr - p_d^2).
//  (M) LINE MISSES.  The reference evaluates det = b*b - 4*a*c in float (contract: no contraction, left to right).  Rounding by
//      rounding -- b's dot product 3 ulps of |d||off|, squared: 7 * 2^-24 of 4 a o2; c: 4 * 2^-24 o2; a: 3 * 2^-24; the product
//      4*a*c and the final subtraction one each -- its error is at most 16 * 2^-24 = 2^-20 of 4 a o2.  So det >= 0 requires
//      p_d^2 - rr <= 2^-20 o2.  Every direction of the footprint is within rho of the centre direction u_c (pt_footprint.h:
//      bilinear directions lie in the hull of the four corner directions; rho = 1.05 max|d_corner - d_c| / |d_c| + 1e-6), and
//      |sin angle(d, off) - sin angle(u_c, off)| <= rho, so p_d >= q := p_c - rho |off|.  The sphere is dropped when
//      q > 0 and q^2 - rr > 1.25 * 2^-18 o2: five times the bound.
//  (B) BEHIND: pt_footprint.h's rule, unchanged: eye robustly outside (c >= 2^-8 o2) and u_c . off > rho |off|: the centre is
//      behind every ray of the footprint, both roots are negative, t > 0 fails.
// The wave first runs the same two tests against ONE cone that contains all its pixels' footprints (64 consecutive columns: a
// strip 50 mrad long), sphere j = lane, lane + 64, ...: 16 trips at 1000 spheres leave ~20 survivors, and each lane then tests
// only those against its own cone.  A wave whose pixels are not one narrow strip (ragged widths), more survivors than the
// scratch list holds, more than kPrimMaxList spheres in a pixel's cone (0.2 % of the pixels of BASELINE's 1000-sphere scene):
// no list -- those pixels walk as before.
//
// SKY.  A pixel whose list is empty in a scene without spheres outside the grid is missed by every primary ray: each sample is
// the reference's `output.color += color; return` with color = 0 (:157-161) after its two jitter draws (:223-224) -- the kernel
// does exactly that, sample by sample, before the sample loop, and the lane enters the loop finished (a helper of its wave).
#pragma once
#include "pt_grid.h"
#include "pt_footprint.h"

#pragma clang fp contract(off)

namespace pt {

#ifndef PT_PRIMLIST
#define PT_PRIMLIST 1  // 0: every primary ray walks the grid (round 4's kernel)
#endif
#ifndef PT_PRIMLIST_MIN_SPP
#define PT_PRIMLIST_MIN_SPP 4  // the lists are built once per pixel and workgroup: worth it from this many samples
#endif
constexpr int kPrimMaxSurvivors = 256;  // of the wave's cone; they sit in the wave's test ring, which is idle before the sample loop
static_assert(kPrimMaxSurvivors <= kPoolRing, "the survivor list lives in the test ring");

// PT_PRIMLIST_MUTANT (never defined in a shipped build) makes the analysis deliberately UNSOUND -- 1: footprint radius x 0.3,
// 2: spheres count as missed when the line passes within 0.8 r -- to show that the soaks notice (tools/primlist_soak.py).
#if defined(PT_PRIMLIST_MUTANT) && PT_PRIMLIST_MUTANT == 1
constexpr float kPrimRadiusFactor = 0.3f;
constexpr double kPrimMissScale = 1.0;
#elif defined(PT_PRIMLIST_MUTANT) && PT_PRIMLIST_MUTANT == 2
constexpr float kPrimRadiusFactor = 1.05f;
constexpr double kPrimMissScale = 0.64;
#else
constexpr float kPrimRadiusFactor = 1.05f;
constexpr double kPrimMissScale = 1.0;
#endif

// can NO direction within `rho` (radians, already inflated) of `axis` (any length) be accepted by the reference for this sphere?
__device__ __forceinline__ bool prim_cone_excludes(const float4 g, F3 eye, F3 axis, float rho) {
  const F3 off = mk3(eye.x - g.x, eye.y - g.y, eye.z - g.z);  // pathtrace.cu:73 with origin = eye: the reference's own float
  const double ox = off.x, oy = off.y, oz = off.z, ux = axis.x, uy = axis.y, uz = axis.z;
  const double o2 = ox * ox + oy * oy + oz * oz, uu = ux * ux + uy * uy + uz * uz, hu = ux * ox + uy * oy + uz * oz;
  const double rr = (double)g.w * kPrimMissScale;
  const double lo = sqrt(o2), lu = sqrt(uu);
  const double p2 = o2 - hu * hu / uu;
  const double q = sqrt(p2 > 0.0 ? p2 : 0.0) - (double)rho * lo;
  const bool miss = (q > 0.0) & (q * q - rr > 4.76837158203125e-06 * o2);       // (M): 1.25 * 2^-18
  const bool behind = (o2 - rr >= 0.00390625 * o2) & (hu > (double)rho * lo * lu * 1.000001);  // (B)
  return miss | behind;
}

// Builds the lane's list (table entries `slot`, `slot + 1` of `cells`) and returns whether it is valid; k_out = its length.
// Every lane of the wave must call (the wave cooperates); `active`: the lane has a pixel.  `ring`: the wave's scratch list.
template <typename DirFn>
__device__ __forceinline__ bool build_primary_list(const GridLds& G, const pt_sphere* __restrict__ spheres, int n, F3 eye,
                                                   float row, float col, bool active, DirFn dir_at, uint32_t* ring, uint2* cells,
                                                   uint32_t slot, int& k_out) {
  const int lane = threadIdx.x & 63;
  k_out = 0;
  // the lane's cone (pt_footprint.h, primary_candidates: same construction)
  const F3 dc = dir_at(row, col);
  const float lc = sqrtf(dot(dc, dc));
  float dev = 0.0f;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const F3 dk = dir_at(row + ((k & 1) ? 0.5f : -0.5f), col + ((k & 2) ? 0.5f : -0.5f));
    const F3 e = dk - dc;
    const float len = sqrtf(dot(e, e));
    dev = !(len <= dev) ? len : dev;  // (a NaN stays)
  }
  const float rho = kPrimRadiusFactor * dev / lc + 1e-6f;
  bool ok = active & (rho <= 0.125f) & (lc > 0.0f) & (dev >= 0.0f);  // NaN or degenerate footprint: no list (1.05 x >= asin x up to 0.3)
  const F3 u = dc * (1.0f / lc);
  // one cone around all the wave's footprints: axis between the first and the last pixel's directions
  const uint64_t okm = __builtin_amdgcn_ballot_w64(ok);
  uint32_t n_surv = 0u;
  bool wave_ok = okm != 0ull;
  if (wave_ok) {
    const int l0 = __builtin_ctzll(okm), l1 = 63 - __builtin_clzll(okm);
    auto rl = [](float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); };
    F3 A = mk3(rl(u.x, l0) + rl(u.x, l1), rl(u.y, l0) + rl(u.y, l1), rl(u.z, l0) + rl(u.z, l1));
    A = A * (1.0f / sqrtf(dot(A, A)));
    const F3 e = u - A;
    // angle(d, A) <= angle(d, u) + angle(u, A), and an angle below 0.5 rad is less than 1.05 of its chord
    float wr = ok ? 1.05f * sqrtf(dot(e, e)) + rho : 0.0f;
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) wr = fmaxf(wr, __shfl_xor(wr, m));
    wave_ok = (wr <= 0.25f) & (wr > 0.0f);  // (wave-uniform; a NaN axis fails both)
    if (wave_ok) {
      for (int j0 = 0; j0 < n; j0 += 64) {
        const int j = j0 + lane;
        bool keep = false;
        if (j < n) {
          const float r = spheres[j].radius;
          const bool in_grid = (r >= G.h.r_small) & (r <= G.h.r_big);  // build_grid_kernel's own compare on the same float
          keep = in_grid && !prim_cone_excludes(G.geom[j], eye, A, wr);
        }
        const uint64_t km = __builtin_amdgcn_ballot_w64(keep);
        const uint32_t pos = n_surv + __builtin_amdgcn_mbcnt_hi((uint32_t)(km >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)km, 0u));
        if (keep && pos < (uint32_t)kPrimMaxSurvivors) ring[pos] = (uint32_t)j;
        n_surv += (uint32_t)__builtin_popcountll(km);
      }
      wave_ok = n_surv <= (uint32_t)kPrimMaxSurvivors;
    }
  }
  ok = ok & wave_ok;
  // the lane's own cone against the survivors; the list as a shift register of 16-bit indices (most recent first)
  unsigned long long lo64 = 0ull;
  uint32_t hi16 = 0u;
  int k = 0;
  if (wave_ok) {
    asm volatile("" ::: "memory");  // (the ring was written by other lanes of this wave: DS operations of one wave execute in order)
    for (uint32_t s = 0; s < n_surv; s++) {
      const uint32_t j = ring[s];
      const bool keep = ok && !prim_cone_excludes(G.geom[j], eye, dc, rho);
      if (keep) {
        hi16 = (uint32_t)(lo64 >> 48);
        lo64 = (lo64 << 16) | j;
        k++;
      }
    }
    asm volatile("" ::: "memory");
  }
  ok = ok & (k <= kPrimMaxList);
  const uint32_t kk = ok ? (uint32_t)k : 0u;
  const uint32_t s0 = (uint32_t)lo64 & 0xFFFFu, s1 = (uint32_t)(lo64 >> 16) & 0xFFFFu, s2 = (uint32_t)(lo64 >> 32) & 0xFFFFu,
                 s3 = (uint32_t)(lo64 >> 48), s4 = hi16 & 0xFFFFu;
  uint2 e0, e1;
  if (kk <= 3u) {  // one entry, no link (pt_grid.h: word 0 = first | link << 16 | count << 30, word 1 = second | third << 16)
    e0 = make_uint2(s0 | (kk << 30), s1 | (s2 << 16));
    e1 = make_uint2(0u, 0u);
  } else {  // two spheres and the link, then the rest
    e0 = make_uint2(s0 | ((slot + 1u) << 16) | (2u << 30), s1);
    e1 = make_uint2(s2 | ((kk - 2u) << 30), s3 | (s4 << 16));
  }
  cells[slot] = e0;
  cells[slot + 1u] = e1;
  k_out = (int)kk;
  return ok;
}

}  // namespace pt
