// pt_grid.h -- variant 11: a conservative uniform grid over the small spheres of a many-sphere scene.
//
// The reference's intersectScene (src/pathtrace.cu:93-107) is brute force and variants 6/8/10 keep it so
// (their sphere loop runs at the VALU issue peak).  This variant changes WHICH spheres a lane tests, never
// what it decides: the grid only has to deliver a SUPERSET of the spheres whose reference `t` can rank
// first or second for the ray; ranking, ambiguity detection, the exact FP64 step and the literal fallback
// are those of intersect_scene_screened_large.  Why the superset property holds:
//  * The reference's float t of sphere j satisfies a t^2 + b t + c = |p - c_j|^2 - r_j^2 with p = o + t d up
//    to the rounding of its own b, c (relative 2^-22 of |off|^2), so the point p it implies lies within
//    2^-21 |off|^2 / r_j of the TRUE sphere surface.  Every sphere is registered in all cells that touch its
//    bounding box inflated by  m_j = 2^-20 D^2 / r_j + slack,  D = 6 E the largest |off| of an admitted ray
//    (E = grid extent; rays whose origin is farther than 5 E from the grid centre take the brute-force loop),
//    so the cell containing p holds j.
//  * The traversal (3D-DDA in float) visits the cells of the real ray in order up to slivers much shorter
//    than `slack` = E 2^-12; a sliver's content is also registered in its neighbours because of the inflation.
//  * A lane stops after a cell only when its best estimate T1 satisfies T1 (1 + 2^-17) < 2a (t_exit - slack_t):
//    every sphere that could beat the winner, or come within the ambiguity margin of it, implies a point in
//    an already visited cell.
//  * Spheres that are too large (r > 8 r_ref: the walls), too small (m_j would explode) or too many for the
//    tables go to a list every lane tests; if the tables overflow the header says invalid and the kernel
//    uses the brute-force loop.
// Evidence: bit-identical to variant 10 and to the CPU oracle on the fuzz and many-sphere tests.
//
// Divergence: a ray tests 12-19 registered spheres on average but the slowest lane of a wave needs ~5x that,
// so the walk runs at ~20 % lane utilisation and the gain over the brute-force loop is 2.2x, not 10x.
// Measured and dropped: running the walk as a state machine inside the regeneration loop (lanes whose walk is
// over wait, are shaded and re-launched in batches of 16-56 while the others keep walking) -- bit-exact, but
// 10-15 % slower at every batch size: the per-trip control and the repeated big shading block cost more
// than the idle lanes did.  Grid resolution (1, 2, 4, 8 cells per sphere) changes the time by < 10 %; a filter
// against re-testing the last two spheres (a sphere spans ~2.9 cells) gained 1 %; resolving ambiguous lanes with the
// reference's FP64 test on a second walk instead of the literal loop over all spheres changed nothing measurable.
#pragma once
#include "pt_intersect.h"

#pragma clang fp contract(off)

namespace pt {

// Grid resolution.  Round 3, with the sphere tests pooled (variant 13) a test costs a 64th of a pass and the balance moved
// towards finer cells: 1000 spheres + walls at 32 spp, cells per sphere 0.5 / 1 / 2 / 3 / 4: 15.3 / 14.0 / 13.9 / 13.6 / 13.0 ms at
// six waves per SIMD, 3: 12.7 and 4: 13.0 at four (profiles/r03/cfg4_ab.txt); variant 11 is indifferent (16.6-16.7 ms).  With sample
// chunking on (steady frame times, 256 spp, closed | open ms): 2: 95.5 | 35.3, 2.25: 96.0 | 35.1, 2.5: 94.4 | 35.3, 2.75: 94.3 | 35.4,
// 3: 95.6 | 35.1, 3.5: 96.3 | 35.4 -- flat within a percent (the steps are the cell counts per axis changing): 2.75.
#ifndef PT_V13_LAST_SHORTCUT
#define PT_V13_LAST_SHORTCUT 1  // grid_begin: a path's last bounce walks only if it may end on an emitting sphere
#endif
// (Measured and dropped, round 5 -- profiles/r05/cfg4_ab.txt block 9: skipping the clip / DDA set-up when no lane of the wave walks, and
// confirming a last bounce's winner from its float part instead of the FP64 step (EXACTNESS.md A.8's rule, in the grid's units): each
// made the kernel 0.5-1 % SLOWER -- a branch and its live ranges cost more than the instructions they skip.  Both were first
// measured as "no effect": their switches were defined BEHIND the code that tested them, so they were silently off.)
#ifndef PT_GRID_MAX_CELLS
#define PT_GRID_MAX_CELLS 4096
#endif
#ifndef PT_GRID_CELLS_PER_SPHERE
#define PT_GRID_CELLS_PER_SPHERE 2.75f
#endif
#ifndef PT_GRID_EXACT_REG
#define PT_GRID_EXACT_REG 1  // register a sphere only in the cells its inflated ball reaches, not in its whole bounding box (build_grid_kernel)
#endif
constexpr int kGridMaxCells = PT_GRID_MAX_CELLS;
#ifndef PT_GRID_TESTS_PER_TRIP
#define PT_GRID_TESTS_PER_TRIP 3
#endif
#ifndef PT_GRID_MAX_ITEMS
#define PT_GRID_MAX_ITEMS 8192
#endif
constexpr int kGridMaxItems = PT_GRID_MAX_ITEMS;
constexpr int kGridMaxBig = 64;
constexpr int kGridMaxSpheres = PT_GRID_MAX_SPHERES;  // geometry of all spheres is staged (16 B each)
constexpr int kGridBuildThreads = 1024;

struct GridHeader {  // 64 bytes, written by build_grid_kernel
  uint32_t valid, nx, ny, nz;
  float ox, oy, oz, cs;
  float inv_cs, slack, cx, cy;
  float cz, far2;  // grid centre and (5E)^2: rays starting farther away are not admitted
  uint32_t n_big, n_items;  // n_big: bits 0-15 = spheres outside the grid, bits 16-31 = entries of the pooled walk's table (cells + chained)
  float r_small, r_big;     // a sphere is registered in the grid iff r_small <= radius <= r_big (pt_primlist.h classifies with the same compare)
  uint32_t prim_base, n_emis; // variant 13: table index of the per-pixel primary-ray lists (behind the room for cells + chained entries);
                              // emitting spheres inside the grid (0xFFFF: more than kGridMaxEmis -- no last-bounce shortcut)
};
static_assert(sizeof(GridHeader) == 80, "GridHeader layout");

// accel buffer: header | big[kGridMaxBig] u16 | cell_start[kGridMaxCells + 2] u16 | items[kGridMaxItems] u16 | cells[kGridMaxEntries] 2 x u32
// `cells` (round 4, variant 13) is the cell table in the form the pooled walk reads with ONE ds_read_b64 per visit.  Entry c
// (c < ncells) belongs to cell c; a cell with more than three registrations continues in chained entries behind the cells:
//   word 0 = first sphere | next << 16 | k << 30     k = spheres in this entry (0..3), next = index of the entry the list
//   word 1 = second sphere | third sphere << 16          continues in (0 = none; then k = 2: two spheres and the link)
// 97 % of the cells of the 1000-sphere scene have at most three registrations and need nothing but their own entry.
constexpr int kGridMaxEntries = 8191;  // 13 bits of `next`
constexpr size_t kGridBigOff = sizeof(GridHeader);
constexpr size_t kGridStartOff = kGridBigOff + kGridMaxBig * sizeof(uint16_t);
constexpr size_t kGridItemsOff = kGridStartOff + (kGridMaxCells + 2) * sizeof(uint16_t);
constexpr size_t kGridCellsOff = (kGridItemsOff + kGridMaxItems * sizeof(uint16_t) + 7) & ~(size_t)7;
// Round 5 (grid_last_shortcut): which spheres EMIT -- a bitmap over all spheres and the list of the emitting spheres inside the grid
constexpr int kGridMaxEmis = 32;                                  // more emitting grid spheres than this: the shortcut is off
constexpr int kGridEmisWords = PT_GRID_MAX_SPHERES / 32 + kGridMaxEmis / 2;  // bitmap, then the list (u16)
constexpr size_t kGridEmisOff = kGridCellsOff + (size_t)(kGridMaxEntries + 1) * 2 * sizeof(uint32_t);
constexpr size_t kGridAccelBytes = kGridEmisOff + (size_t)kGridEmisWords * sizeof(uint32_t);

// How many table entries the pooled kernel's grid may have (cells + chained entries).  Variant 11's image has room for the maxima
// of its own tables; variant 13's (8 bytes per entry, no registration list, plus 2.75 KB of test pool per wave) is sized so that
// TWO 512-thread workgroups share a CU's 160 KB of LDS whatever the scene size (1000 spheres: 4 666 entries -- the build uses
// 3 003 cells + about 100 chained; 2048 spheres: 2 570).
constexpr int kPoolLdsTarget = 80 * 1024;  // per workgroup
constexpr int kPoolRing = 512;             // ring entries per wave: a round adds at most 64 * 3 * PT_POOL_STEPS to fewer than 64 pending
constexpr int kPoolWaveBytes = kPoolRing * 4 + 64 * 8 + 64 * 4 + 64 * 4 + 16;  // ring, best keys, runner-up estimates, the sweep's owner list + slot mask
constexpr int kGridBigGeomBytes = kGridMaxBig * (int)sizeof(float4);
// Round 5: behind the cells and their chained entries the table holds, per lane of the workgroup, the list of grid spheres the
// lane's pixel can see along a PRIMARY ray (pt_primlist.h), in the table's own entry format -- two entries: up to five spheres --
// so that bounce 0 feeds the pooled tests from it and never walks.  Link fields are 13 bits: the lists must end below 8192.
constexpr int kPrimEntriesPerLane = 2;
constexpr int kPrimMaxList = 5;  // 2 spheres + link, then up to 3
// `threads`: the workgroup size of the kernel that stages the image -- PT_GRID_BLOCK_THREADS (512: two workgroups share a CU's LDS),
// or PT_GRID_WIDE_THREADS (1024, "variant 14": ONE workgroup per CU, the same four waves per SIMD, one image instead of two -- the
// cell table gets the other half of the LDS, which scenes above ~1200 spheres need: at 2048 spheres 1 546 entries become 7 000)
__host__ __device__ inline int grid_prim_entries(int threads) { return threads * kPrimEntriesPerLane; }
__host__ __device__ inline int grid_lds_target(int threads) { return threads > PT_GRID_BLOCK_THREADS ? 2 * kPoolLdsTarget - 2048 : kPoolLdsTarget; }
__host__ __device__ inline int grid_max_entries(int n, bool pooled, int threads = PT_GRID_BLOCK_THREADS) {
  if (!pooled) return kGridMaxEntries;  // (variant 11 does not stage the table)
  const int prim = grid_prim_entries(threads);
  const int fixed = kTablesF4 * (int)sizeof(float4) + (threads / 64) * kPoolWaveBytes + kGridBigGeomBytes +
                    kGridMaxBig * (int)sizeof(uint16_t) + 32 + prim * 8 + kGridEmisWords * 4;
  int avail = grid_lds_target(threads) - fixed - n * (int)sizeof(float4);
  if (avail < 8192) avail = 8192;  // (never with n <= kGridMaxSpheres)
  const int e = avail / 8;
  return e > kGridMaxEntries - prim ? kGridMaxEntries - prim : e;
}

// LDS image per workgroup: geometry of all n spheres, the geometry of the spheres outside the grid once more (contiguous: no
// index read on the way to it), then the tables (dword aligned)
__host__ __device__ inline size_t grid_lds_bytes(int n, bool pooled, int threads = PT_GRID_BLOCK_THREADS) {
  const size_t head = (size_t)n * sizeof(float4) + kGridBigGeomBytes + kGridMaxBig * sizeof(uint16_t);
  if (pooled) return head + (size_t)(grid_max_entries(n, true, threads) + grid_prim_entries(threads)) * 8 + (size_t)kGridEmisWords * 4;
  return head + (kGridCellsOff - kGridStartOff);
}

struct GridLds {
  bool valid;  // wave-uniform
  GridHeader h;
  const float4* geom;
  const float4* bigg;  // geometry of the spheres outside the grid, in the order of `big`
  const uint16_t* big;
  const uint16_t* cell_start;  // variants 11, 12
  const uint16_t* items;
  const uint2* cells;          // variant 13
  const uint32_t* em_bits;     // variant 13: bit i set = sphere i emits (any emission component other than +-0)
  const uint16_t* em_list;     // ... and the emitting spheres inside the grid (GridHeader::n_emis of them)
};

template <bool POOLED>
__device__ __forceinline__ GridLds stage_grid(const pt_sphere* __restrict__ spheres, int n, const uint32_t* __restrict__ accel,
                                              float4* lds) {
  GridLds g;
  g.h = *reinterpret_cast<const GridHeader*>(accel);
  g.valid = g.h.valid != 0u;
  g.geom = lds;
  float4* bigg = lds + n;
  g.bigg = bigg;
  uint32_t* tab = reinterpret_cast<uint32_t*>(bigg + kGridMaxBig);
  g.big = reinterpret_cast<const uint16_t*>(tab);
  const int ncells = (int)(g.h.nx * g.h.ny * g.h.nz);
  uint32_t* after_big = tab + kGridMaxBig / 2;
  g.em_bits = nullptr;
  g.em_list = nullptr;
  uint32_t* em_lds = nullptr;
  if constexpr (POOLED) {
    g.cells = reinterpret_cast<const uint2*>(after_big);  // (8-byte aligned: everything before it is a multiple of 16 bytes)
    g.items = nullptr;  // not staged: the table's chained entries carry the long lists
    g.cell_start = nullptr;
    // behind the cells and the per-lane primary lists (grid_lds_bytes): the emission bitmap and list
    em_lds = after_big + 2 * (size_t)(grid_max_entries(n, true, (int)blockDim.x) + grid_prim_entries((int)blockDim.x));
    g.em_bits = em_lds;
    g.em_list = reinterpret_cast<const uint16_t*>(em_lds + PT_GRID_MAX_SPHERES / 32);
  } else {
    g.cell_start = reinterpret_cast<const uint16_t*>(after_big);
    g.items = g.cell_start + (kGridMaxCells + 2);
    g.cells = nullptr;
  }
  if (!g.valid) return g;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const pt_sphere sp = spheres[i];
    lds[i] = make_float4(sp.pos[0], sp.pos[1], sp.pos[2], sp.radius * sp.radius);
  }
  const int n_big = (int)(g.h.n_big & 0xFFFFu);
  for (int i = threadIdx.x; i < n_big; i += blockDim.x) {
    const pt_sphere sp = spheres[reinterpret_cast<const uint16_t*>(accel + kGridBigOff / 4)[i]];
    bigg[i] = make_float4(sp.pos[0], sp.pos[1], sp.pos[2], sp.radius * sp.radius);
  }
  const uint32_t* src = accel + kGridBigOff / 4;
  for (int i = threadIdx.x; i < kGridMaxBig / 2; i += blockDim.x) tab[i] = src[i];
  if constexpr (POOLED) {
    const int n_entries = (int)(g.h.n_big >> 16);
    const uint32_t* src_cells = accel + kGridCellsOff / 4;
    for (int i = threadIdx.x; i < 2 * n_entries; i += blockDim.x) after_big[i] = src_cells[i];
    const uint32_t* src_em = accel + kGridEmisOff / 4;
    for (int i = threadIdx.x; i < kGridEmisWords; i += blockDim.x) em_lds[i] = src_em[i];
  } else {
    const int words_i = ((int)g.h.n_items + 1) / 2;
    const uint32_t* src_items = accel + kGridItemsOff / 4;
    const uint32_t* src_start = accel + kGridStartOff / 4;
    const int words_s = (ncells + 2 + 1) / 2;
    for (int i = threadIdx.x; i < words_s; i += blockDim.x) after_big[i] = src_start[i];
    uint32_t* dst_items = after_big + (kGridMaxCells + 2) / 2;
    for (int i = threadIdx.x; i < words_i; i += blockDim.x) dst_items[i] = src_items[i];
  }
  __syncthreads();
  return g;
}

// ---- build: one workgroup, everything in LDS -------------------------------------------------------
__device__ __forceinline__ int f2ord(float f) {  // order-preserving float -> int
  const int i = __float_as_int(f);
  return i >= 0 ? i : (int)(0x80000000u - (uint32_t)i);
}
__device__ __forceinline__ float ord2f(int i) { return __int_as_float(i >= 0 ? i : (int)(0x80000000u - (uint32_t)i)); }

// entries chained behind a cell's own one in the pooled walk's table: a cell with count <= 3 needs none; otherwise every entry but
// the last holds two spheres and the link, the last up to three
__host__ __device__ inline uint32_t chained_entries(uint32_t count) { return count <= 3u ? 0u : (count - 2u) / 2u; }

// eye_valid != 0: the camera position of the frame.  It only sizes the ADMISSION radius (header far2): every ray is checked
// against far2 before it may use the grid, and the registration margins below are derived from that same radius, so a
// wrong or missing hint costs time (more rays on the brute-force path, or fatter registrations), never correctness.
__global__ void __launch_bounds__(kGridBuildThreads) build_grid_kernel(const pt_sphere* __restrict__ spheres, int n,
                                                                         uint32_t* __restrict__ accel, float eye_x, float eye_y,
                                                                         float eye_z, int eye_valid, int max_entries) {
  __shared__ uint32_t cnt[kGridMaxCells + 1];
  __shared__ uint32_t scan_tmp[kGridBuildThreads];
  __shared__ int bb[6];
  __shared__ float fsum;
  __shared__ uint32_t n_small, n_big, total, total_ext, s_n_emis;
  __shared__ float s_cs;
  __shared__ uint32_t s_dims[3];
  __shared__ float enc[6];
  GridHeader* hdr = reinterpret_cast<GridHeader*>(accel);
  uint16_t* big = reinterpret_cast<uint16_t*>(reinterpret_cast<char*>(accel) + kGridBigOff);
  uint16_t* cell_start = reinterpret_cast<uint16_t*>(reinterpret_cast<char*>(accel) + kGridStartOff);
  uint16_t* items = reinterpret_cast<uint16_t*>(reinterpret_cast<char*>(accel) + kGridItemsOff);
  const int tid = threadIdx.x;
  auto invalid = [&]() {
    if (tid == 0) hdr->valid = 0u;
  };
  if (n > kGridMaxSpheres || n < 1) {
    invalid();
    return;
  }
  max_entries = max_entries < kGridMaxEntries ? max_entries : kGridMaxEntries;
  // reference radius: geometric mean (a handful of huge walls hardly move it)
  if (tid == 0) {
    fsum = 0.0f;
    n_small = n_big = total = 0u;
    s_n_emis = 0u;
    bb[0] = bb[1] = bb[2] = 0x7FFFFFFF;
    bb[3] = bb[4] = bb[5] = (int)0x80000000;
  }
  __syncthreads();
  float lsum = 0.0f;
  for (int i = tid; i < n; i += kGridBuildThreads) lsum += log2f(fmaxf(spheres[i].radius, 1e-30f));
  atomicAdd(&fsum, lsum);
  __syncthreads();
  const float r_ref = exp2f(fsum / (float)n);
  const float r_big = 8.0f * r_ref;
  // bounding box of the candidate spheres (not too large)
  for (int i = tid; i < n; i += kGridBuildThreads) {
    const pt_sphere sp = spheres[i];
    if (sp.radius > 0.0f && sp.radius <= r_big) {
      for (int k = 0; k < 3; k++) {
        atomicMin(&bb[k], f2ord(sp.pos[k] - sp.radius));
        atomicMax(&bb[3 + k], f2ord(sp.pos[k] + sp.radius));
      }
    }
  }
  __syncthreads();
  if (bb[0] == 0x7FFFFFFF) {  // nothing to put in a grid
    invalid();
    return;
  }
  float lo[3], hi[3];
  for (int k = 0; k < 3; k++) {
    lo[k] = ord2f(bb[k]);
    hi[k] = ord2f(bb[3 + k]);
  }
  const float E = fmaxf(fmaxf(hi[0] - lo[0], hi[1] - lo[1]), hi[2] - lo[2]);
  if (!(E > 0.0f) || !(E < 1e15f)) {
    invalid();
    return;
  }
  const float slack = E * 0.0001220703125f;         // 2^-13 E (the float DDA is off by ~2^-20 of the coordinates after 100 steps)
  // Admission radius `far` (rays starting farther from the box centre take the brute-force loop) and D, the largest
  // |origin - sphere centre| an admitted ray can have: origins of secondary rays lie on the scene (inside the box or on
  // the walls around it: within E of the centre), the primary rays start at the eye.  m_j = 2^-20 D^2 / r_j + slack.
  const float ctr[3] = {0.5f * (lo[0] + hi[0]), 0.5f * (lo[1] + hi[1]), 0.5f * (lo[2] + hi[2])};
  // Where can a secondary ray start?  On a grid sphere (inside the box) or on one of the big spheres around it (the
  // walls).  Six axis rays from the box centre against the big spheres measure that enclosure (an estimate: it only has
  // to be good enough that few rays start beyond `far`); a direction without a big sphere counts the box face.
  if (tid < 6) {
    const int ax = tid >> 1;
    const float sg = (tid & 1) ? -1.0f : 1.0f;
    float best = 3.0e38f;
    for (int i = 0; i < n; i++) {
      const pt_sphere sp = spheres[i];
      if (!(sp.radius > r_big)) continue;
      const double ox = (double)ctr[0] - sp.pos[0], oy = (double)ctr[1] - sp.pos[1], oz = (double)ctr[2] - sp.pos[2];
      const double hb = sg * (ax == 0 ? ox : (ax == 1 ? oy : oz));  // dot(dir, off), |dir| = 1
      const double cc = ox * ox + oy * oy + oz * oz - (double)sp.radius * sp.radius;
      const double disc = hb * hb - cc;
      if (disc < 0.0) continue;
      const double sq = sqrt(disc), t0 = -hb - sq, t1 = -hb + sq;
      const double t = t0 > 0.0 ? t0 : t1;
      if (t > 0.0 && t < (double)best) best = (float)t;
    }
    const float face = 0.5f * (hi[ax] - lo[ax]);
    enc[tid] = best < 3.0e38f ? best : face;
  }
  __syncthreads();
  const float enx = fmaxf(enc[0], enc[1]), eny = fmaxf(enc[2], enc[3]), enz = fmaxf(enc[4], enc[5]);
  float far = fminf(fmaxf(1.03125f * sqrtf(enx * enx + eny * eny + enz * enz), E), 5.0f * E);  // never beyond the old fixed 5 E
  if (eye_valid) {
    const float ex = eye_x - ctr[0], ey = eye_y - ctr[1], ez = eye_z - ctr[2];
    const float de = sqrtf(ex * ex + ey * ey + ez * ez);
    if (de == de && de < 1e15f) far = fmaxf(far, 1.0625f * de);  // the primary rays
  } else {
    far = 5.0f * E;  // without a camera hint: generous
  }
  const float D = far + 0.875f * E;                 // + half the box diagonal (<= sqrt(3)/2 E)
  const float mk = D * D * 9.5367431640625e-07f;    // 2^-20 D^2
  const float r_small = E * 0.001953125f;           // below 2^-9 E the inflation would dwarf the sphere
  auto margin = [&](float r) { return mk / r + slack; };
  auto in_grid = [&](float r) { return r >= r_small && r <= r_big; };
  // Round 5: a sphere is registered in the cells of its inflated bounding box that its inflated BALL reaches (the corner cells of a
  // 2 x 2 x 2 footprint mostly are not reached: 8-14 % fewer registrations, as many fewer tests).  EXACTNESS A.6 (i)-(ii) need the
  // cell of every point within r + 2^-21 D^2 / r of the centre, and that cell's neighbours for points within `slack` of a face:
  // the ball of radius m + slack (m = r + margin(r), margin = 2^-20 D^2 / r + slack) covers both; `guard` pays for the rounding of
  // the cell faces lo + x * cs where the scene lies far from the origin.
  float guard = 0.0f;
  for (int k = 0; k < 3; k++) guard = fmaxf(guard, fmaxf(fabsf(lo[k]), fabsf(hi[k])));
  guard *= 4.76837158203125e-07f;  // 2^-21: four ulps of the largest coordinate
  auto reaches = [&](const float rel[3], float reach, float cell, int x, int y, int z) {
#if PT_GRID_EXACT_REG
    const int c[3] = {x, y, z};
    float d2 = 0.0f;
    for (int k = 0; k < 3; k++) {
      const float f0 = (float)c[k] * cell, f1 = (float)(c[k] + 1) * cell;
      const float dk = fmaxf(fmaxf(f0 - rel[k], rel[k] - f1), 0.0f);
      d2 += dk * dk;
    }
#if defined(PT_GRID_REG_MUTANT)  // soak self-test: a ball that is too small must be caught (profiles/r05/grid_reg.txt)
    reach = reach * 0.93f;
#endif
    return d2 <= reach * reach * 1.00001f;
#else
    return true;
#endif
  };
  // grid box = bounding box inflated by the largest margin that can occur
  const float m_max = margin(r_small);
  for (int k = 0; k < 3; k++) {
    lo[k] -= m_max + slack;
    hi[k] += m_max + slack;
  }
  // count the candidates, pick a cell size: about two cells per sphere, at least the reference diameter
  uint32_t ls = 0;
  for (int i = tid; i < n; i += kGridBuildThreads) ls += in_grid(spheres[i].radius) ? 1u : 0u;
  atomicAdd(&n_small, ls);
  __syncthreads();
  if (n_small == 0u || n - (int)n_small > kGridMaxBig) {
    invalid();
    return;
  }
  const float sx = hi[0] - lo[0], sy = hi[1] - lo[1], sz = hi[2] - lo[2];
  float cs = fmaxf(cbrtf(sx * sy * sz / (PT_GRID_CELLS_PER_SPHERE * (float)n_small)), 2.0f * r_ref);
  for (int attempt = 0; attempt < 12; attempt++) {
    // at most 512 cells along one axis: the traversal keeps its three remaining-cell counters in one register (10 bits each)
    const uint32_t nx = (uint32_t)fminf(ceilf(sx / cs), 512.0f), ny = (uint32_t)fminf(ceilf(sy / cs), 512.0f),
                   nz = (uint32_t)fminf(ceilf(sz / cs), 512.0f);
    const uint32_t ncells = (nx < 1 ? 1 : nx) * (ny < 1 ? 1 : ny) * (nz < 1 ? 1 : nz);
    bool ok = ncells <= (uint32_t)kGridMaxCells && ncells <= (uint32_t)max_entries;
    if (ok) {
      // count the registrations at this cell size
      __syncthreads();
      for (int c = tid; c <= kGridMaxCells; c += kGridBuildThreads) cnt[c] = 0u;
      if (tid == 0) total = 0u, total_ext = 0u;
      __syncthreads();
      const float inv = 1.0f / cs;
      uint32_t lt = 0;
      for (int i = tid; i < n; i += kGridBuildThreads) {
        const pt_sphere sp = spheres[i];
        if (!in_grid(sp.radius)) continue;
        const float m = sp.radius + margin(sp.radius);
        int a[3], b[3];
        const uint32_t dims[3] = {nx, ny, nz};
        for (int k = 0; k < 3; k++) {
          a[k] = (int)floorf((sp.pos[k] - m - lo[k]) * inv);
          b[k] = (int)floorf((sp.pos[k] + m - lo[k]) * inv);
          a[k] = a[k] < 0 ? 0 : a[k];
          b[k] = b[k] >= (int)dims[k] ? (int)dims[k] - 1 : b[k];
        }
        const float rel[3] = {sp.pos[0] - lo[0], sp.pos[1] - lo[1], sp.pos[2] - lo[2]};
        const float reach = m + slack + guard;
        for (int z = a[2]; z <= b[2]; z++)
          for (int y = a[1]; y <= b[1]; y++)
            for (int x = a[0]; x <= b[0]; x++) {
              if (!reaches(rel, reach, cs, x, y, z)) continue;
              atomicAdd(&cnt[(z * (int)ny + y) * (int)nx + x], 1u);
              lt++;
            }
      }
      atomicAdd(&total, lt);
      __syncthreads();
      ok = total <= (uint32_t)kGridMaxItems;
      if (ok) {  // ... and the chained entries of the pooled walk's table (cells with more than three registrations)
        uint32_t le = 0;
        for (int c = tid; c < (int)ncells; c += kGridBuildThreads) le += chained_entries(cnt[c]);
        atomicAdd(&total_ext, le);
        __syncthreads();
        ok = ncells + total_ext <= (uint32_t)max_entries;
      }
      if (ok && tid == 0) {
        s_cs = cs;
        s_dims[0] = nx;
        s_dims[1] = ny;
        s_dims[2] = nz;
      }
    }
    __syncthreads();
    if (ok) break;
    cs *= 1.26f;  // coarser: half the cells
    if (attempt == 11) {
      invalid();
      return;
    }
  }
  __syncthreads();
  cs = s_cs;
  const uint32_t nx = s_dims[0], ny = s_dims[1], nz = s_dims[2];
  const uint32_t ncells = nx * ny * nz;
  // exclusive scan of cnt[0..ncells): kPer consecutive cells per thread
  {
    constexpr int kPer = (kGridMaxCells + kGridBuildThreads - 1) / kGridBuildThreads;
    uint32_t v[kPer], sum = 0u;
#pragma unroll
    for (int j = 0; j < kPer; j++) {
      const int c = kPer * tid + j;
      v[j] = c < (int)ncells ? cnt[c] : 0u;
      sum += v[j];
    }
    scan_tmp[tid] = sum;
    __syncthreads();
    for (int off = 1; off < kGridBuildThreads; off <<= 1) {
      const uint32_t u = tid >= off ? scan_tmp[tid - off] : 0u;
      __syncthreads();
      scan_tmp[tid] += u;
      __syncthreads();
    }
    uint32_t run = scan_tmp[tid] - sum;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kPer; j++) {
      const int c = kPer * tid + j;
      if (c < (int)ncells) cnt[c] = run;
      run += v[j];
    }
    __syncthreads();
  }
  for (int c = tid; c < (int)ncells; c += kGridBuildThreads) cell_start[c] = (uint16_t)cnt[c];
  if (tid == 0) {
    cell_start[ncells] = (uint16_t)total;
    cell_start[ncells + 1] = (uint16_t)total;
  }
  __syncthreads();
  // fill (cnt[] now serves as the per-cell cursor); everything else goes to the list every lane tests
  const float inv = 1.0f / cs;
  for (int i = tid; i < n; i += kGridBuildThreads) {
    const pt_sphere sp = spheres[i];
    if (!in_grid(sp.radius)) {
      const uint32_t p = atomicAdd(&n_big, 1u);
      if (p < (uint32_t)kGridMaxBig) big[p] = (uint16_t)i;
      continue;
    }
    const float m = sp.radius + margin(sp.radius);
    int a[3], b[3];
    const uint32_t dims[3] = {nx, ny, nz};
    for (int k = 0; k < 3; k++) {
      a[k] = (int)floorf((sp.pos[k] - m - lo[k]) * inv);
      b[k] = (int)floorf((sp.pos[k] + m - lo[k]) * inv);
      a[k] = a[k] < 0 ? 0 : a[k];
      b[k] = b[k] >= (int)dims[k] ? (int)dims[k] - 1 : b[k];
    }
    const float rel[3] = {sp.pos[0] - lo[0], sp.pos[1] - lo[1], sp.pos[2] - lo[2]};
    const float reach = m + slack + guard;
    for (int z = a[2]; z <= b[2]; z++)
      for (int y = a[1]; y <= b[1]; y++)
        for (int x = a[0]; x <= b[0]; x++) {
          if (!reaches(rel, reach, cs, x, y, z)) continue;  // (the count above took the same decision: same operands)
          const uint32_t p = atomicAdd(&cnt[(z * (int)ny + y) * (int)nx + x], 1u);
          items[p] = (uint16_t)i;
        }
  }
  // which spheres emit (grid_last_shortcut): bitmap over all spheres, list of the emitting ones inside the grid
  {
    uint32_t* em = accel + kGridEmisOff / 4;
    uint16_t* em_list = reinterpret_cast<uint16_t*>(em + PT_GRID_MAX_SPHERES / 32);
    for (int wd = tid; wd < PT_GRID_MAX_SPHERES / 32; wd += kGridBuildThreads) {
      uint32_t bits = 0u;
      for (int b = 0; b < 32; b++) {
        const int i = 32 * wd + b;
        if (i < n) {
          const pt_sphere sp = spheres[i];
          const uint32_t any = (__float_as_uint(sp.emission[0]) | __float_as_uint(sp.emission[1]) | __float_as_uint(sp.emission[2])) & 0x7FFFFFFFu;
          if (any != 0u) {  // (a NaN or an infinity emits, too)
            bits |= 1u << b;
            if (in_grid(sp.radius)) {
              const uint32_t p = atomicAdd(&s_n_emis, 1u);
              if (p < (uint32_t)kGridMaxEmis) em_list[p] = (uint16_t)i;
            }
          }
        }
      }
      em[wd] = bits;
    }
  }
  __threadfence();
  __syncthreads();
  // The cell table in the pooled walk's form (layout: head of this file).  Where a cell's chained entries start: exclusive scan
  // of chained_entries(count) over the cells, with the scan code above (cnt[] is free again: the fill is over).
  uint32_t n_ext = 0u;
  {
    constexpr int kPer = (kGridMaxCells + kGridBuildThreads - 1) / kGridBuildThreads;
    uint32_t v[kPer], sum = 0u;
#pragma unroll
    for (int j = 0; j < kPer; j++) {
      const int c = kPer * tid + j;
      v[j] = c < (int)ncells ? chained_entries((uint32_t)cell_start[c + 1] - (uint32_t)cell_start[c]) : 0u;
      sum += v[j];
    }
    __syncthreads();
    scan_tmp[tid] = sum;
    __syncthreads();
    for (int off = 1; off < kGridBuildThreads; off <<= 1) {
      const uint32_t u = tid >= off ? scan_tmp[tid - off] : 0u;
      __syncthreads();
      scan_tmp[tid] += u;
      __syncthreads();
    }
    uint32_t run = scan_tmp[tid] - sum;
    n_ext = scan_tmp[kGridBuildThreads - 1];
    uint32_t* cells = accel + kGridCellsOff / 4;
#pragma unroll
    for (int j = 0; j < kPer; j++) {
      const int c = kPer * tid + j;
      if (c < (int)ncells) {
        uint32_t p = cell_start[c], rem = (uint32_t)cell_start[c + 1] - p, idx = (uint32_t)c, next = ncells + run;
        for (;;) {
          const uint32_t k = rem <= 3u ? rem : 2u, link = rem <= 3u ? 0u : next;
          const uint32_t i0 = k > 0u ? items[p] : 0u, i1 = k > 1u ? items[p + 1u] : 0u, i2 = k > 2u ? items[p + 2u] : 0u;
          cells[2 * idx] = i0 | (link << 16) | (k << 30);
          cells[2 * idx + 1] = i1 | (i2 << 16);
          if (link == 0u) break;
          idx = link;
          next++;
          p += 2u;
          rem -= 2u;
        }
      }
      run += v[j];
    }
  }
  if (tid == 0) {
    GridHeader h;
    h.valid = 1u;
    h.nx = nx;
    h.ny = ny;
    h.nz = nz;
    h.ox = lo[0];
    h.oy = lo[1];
    h.oz = lo[2];
    h.cs = cs;
    h.inv_cs = inv;
    h.slack = slack;
    h.cx = ctr[0];
    h.cy = ctr[1];
    h.cz = ctr[2];
    h.far2 = far * far;
    h.n_big = n_big | ((ncells + n_ext) << 16);
    h.n_items = total;
    h.r_small = r_small;
    h.r_big = r_big;
    h.prim_base = (uint32_t)max_entries;
    h.n_emis = s_n_emis > (uint32_t)kGridMaxEmis ? 0xFFFFu : s_n_emis;
    *hdr = h;
  }
}

// ---- traversal ---------------------------------------------------------------------------------------
#ifdef PT_GRID_STATS  // instrumentation build only (tools/grid_stats.py): what the walk loop does per wave
// [0] walks (wave level), [1] test trips, [2] lanes testing summed over trips, [3] step rounds, [4] lanes stepping summed
// over rounds, [5] lanes that entered the walk, [6] ambiguous lanes (literal loop)
__device__ unsigned long long g_grid_stats[8];
// histograms: [0] sphere tests per ray, [1] cell steps per ray, [2] test trips per wave walk, [3] step rounds per wave walk (bucket 63 = 63 and more)
__device__ unsigned long long g_grid_hist[4][64];
#define PT_HIST_DECL int hist_tests = 0, hist_steps = 0, hist_trips = 0, hist_rounds = 0
#define PT_HIST_LANE(var, n) do { var += (n); } while (0)
#define PT_HIST_END(in_walk) do { \
    if (in_walk) { atomicAdd(&g_grid_hist[0][hist_tests < 63 ? hist_tests : 63], 1ull); atomicAdd(&g_grid_hist[1][hist_steps < 63 ? hist_steps : 63], 1ull); } \
    if ((threadIdx.x & 63) == __builtin_ctzll(__builtin_amdgcn_ballot_w64(true))) { \
      atomicAdd(&g_grid_hist[2][hist_trips < 63 ? hist_trips : 63], 1ull); atomicAdd(&g_grid_hist[3][hist_rounds < 63 ? hist_rounds : 63], 1ull); } } while (0)
#define PT_STAT(i, v) do { const unsigned long long v_ = (unsigned long long)(v); /* evaluated by the whole wave */ \
    if ((threadIdx.x & 63) == __builtin_ctzll(__builtin_amdgcn_ballot_w64(true))) atomicAdd(&g_grid_stats[i], v_); } while (0)
#else
#define PT_STAT(i, v) do { } while (0)
#define PT_HIST_DECL
#define PT_HIST_LANE(var, n) do { } while (0)
#define PT_HIST_END(in_walk) do { } while (0)
#endif
#ifdef PT_GRID_STATS_AMBIG
#define PT_STATW(i, v) do { } while (0)   // the walk's own counters make room for the reasons
#else
#define PT_STATW(i, v) PT_STAT(i, v)
#endif
struct Near2 {
  float T1, T2;  // two smallest estimates of 2a*t
  int i1;
};

// One sphere for one lane: the float part and the estimate of intersect_scene_screened_large, predicated.
// Returns "doubt": the estimate cannot be trusted (its numerator has lost its leading digits: the ray starts within
// rounding distance of the sphere's surface; pt_intersect.h, screen_sphere_oc).  Such a sphere has NOT been entered:
// near2_exact() decides it with the reference's own expression.  (Until round 2 a doubt anywhere sent the lane to the
// literal loop over ALL spheres of the scene: 5e-4 of the lanes per bounce at 1000 spheres -- one lane in 2.4 % of the
// wave-bounces -- and with it a quarter of the frame time in the closed and 40 % in the open configuration.)
// ONCE: the caller meets every sphere at most once (the list of spheres outside the grid): no leader check.
template <bool ONCE = false>
__device__ __forceinline__ bool near2_test(Near2& s, const float4 g, int i, F3 o, F3 d, float a4, float Tlim_hi) {
  const float INF = __builtin_inff();
  const F3 off = mk3(o.x - g.x, o.y - g.y, o.z - g.z);
  const float b = 2.0f * dot(d, off);
  const float c = dot(off, off) - g.w;
  const float bb = b * b;
  const float a4c = a4 * c;
  const float dacc = fmaf(-a4, c, bb);
  // a sphere can sit in several cells: the current leader must not be entered again as its own runner-up
  const bool cand = ((int)__float_as_uint(dacc) >= 0) & (ONCE || !((i == s.i1) & (s.T1 < INF)));
  const float sq = __builtin_amdgcn_sqrtf(dacc);
  const float e = fmaf(b, b, -bb);
  const float num = a4c + e;
  const float TA = copysign_neg_b3(sq, b) - b;  // -q, q = b + copysign(sq, b) (screen_sphere_oc)
  const float TB = num * __builtin_amdgcn_rcpf(TA);
  // the root the reference returns -- the smaller positive one -- is the unsigned minimum of the two bit patterns, and a word
  // with the sign bit set (or a NaN's bits) when there is none (screen_sphere_oc, pt_intersect.h): one unsigned compare with the
  // limit's bits is "positive, finite and below the limit"
  const uint32_t ta = __float_as_uint(TA), tb = __float_as_uint(TB);
  const uint32_t tbits = ta < tb ? ta : tb;
  const float T = __uint_as_float(tbits);
  const bool sure = fabsf(a4c) > fmaf(bb, 1.1920929e-07f, 1e-30f);  // num = a4c + e keeps its leading digits (screen_sphere_oc)
  const bool ok = cand & sure & (tbits < __float_as_uint(Tlim_hi));
  const float Te = ok ? T : INF;
  s.i1 = Te < s.T1 ? i : s.i1;
  s.T2 = __builtin_amdgcn_fmed3f(s.T1, s.T2, Te);  // the second smallest of {T1, T2, Te} (T1 <= T2 always)
  s.T1 = fminf(s.T1, Te);
  return cand & !sure;
}

// a sphere near2_test had doubts about: the reference's own test (pathtrace.cu:72-91, FP64 sqrt and divides) and its
// acceptance rule (:99: t > 0; the upper limit is applied to the winner).  2a*t is entered in place of the estimate -- it
// is the quantity the estimates approximate, 2^-24 away from it, far inside the margins the ranking keeps.
__device__ __forceinline__ void near2_exact(Near2& s, const float4 g, int i, F3 o, F3 d, float a, float Tlim_hi) {
  const float INF = __builtin_inff();
  float t = 0.0f;
  const bool h = intersect_sphere(o, d, a, g, t);
  const float T = (2.0f * a) * t;
  const bool ok = h & (t > 0.0f) & (T < Tlim_hi) & !((i == s.i1) & (s.T1 < INF));
  const float Te = ok ? T : INF;
  s.i1 = Te < s.T1 ? i : s.i1;
  s.T2 = __builtin_amdgcn_fmed3f(s.T1, s.T2, Te);
  s.T1 = fminf(s.T1, Te);
}

// up to three doubted spheres of one trip (bit k of `doubts`: the k-th of them), one copy of the FP64 test
__device__ __forceinline__ void near2_resolve(Near2& s, uint32_t doubts, const float4 g0, int i0, const float4 g1, int i1,
                                              const float4 g2, int i2, F3 o, F3 d, float a, float Tlim_hi) {
  while (doubts != 0u) {
    const bool k0 = (doubts & 1u) != 0u, k1 = !k0 & ((doubts & 2u) != 0u);
    const float4 g = k0 ? g0 : (k1 ? g1 : g2);
    const int i = k0 ? i0 : (k1 ? i1 : i2);
    near2_exact(s, g, i, o, d, a, Tlim_hi);
    doubts &= k0 ? ~1u : (k1 ? ~2u : ~4u);
  }
}

// Nearest hit through the grid.  Loop shape ("test-major"): every trip of the main loop tests ONE registered sphere for
// every lane that is still walking; a lane that has used up its cell's list moves on -- in the small inner loop, which runs
// until every walking lane stands in a non-empty cell again (or has left the grid / met its stop condition).  A lane's trip
// count is its number of sphere tests, not tests + cells, and a wave never waits for the longest CELL LIST of the moment
// (the previous shape: one cell per trip, an inner loop over the longest list among the 64 lanes -- 6 750 VALU instructions
// per wave and bounce at 1000 spheres, a third of the lanes testing at any time).
// One step costs ~22 instructions: the exit face is the smallest tmax; the linear cell index moves by a per-ray stride and
// the three "cells left before the box ends" counters sit in one register, 10 bits each with a guard bit on top of each
// field (a decrement that clears a guard bit has left the box), so no per-axis cell coordinates or bounds tests are kept.
// (The ray's FP64 constants are formed AFTER the walk: six registers the loop does not have to carry.)
//
// The walk is three pieces -- grid_begin (the spheres outside the grid, the clip against the box, the DDA state),
// grid_trips (the loop) and grid_end (the exact step on the winner) -- so that variant 12 can keep a lane's GridWalk alive
// across loop iterations of the pixel kernel; variant 11 calls them back to back.
struct GridWalk {
  Near2 s;
  float a;                         // dot(d, d)
  float tmax0, tmax1, tmax2, tdel0, tdel1, tdel2;
  int cidx;
  int cs0, cs1, cs2;               // what one step along the axis adds to the linear cell index (three scalars, NOT an array:
                                   // a select between array elements becomes an indexed load from scratch memory)
  uint32_t left;                   // per axis: cells left before the box ends, bits 10k..10k+8, guard bit 10k+9
  uint32_t k0, k1;                 // the list of the cell being tested
  uint32_t n0, n1;                 // the list of the NEXT cell, fetched ahead of need (valid while have_next)
  bool have_next;
  bool walking;                    // the DDA can still move on: neither stopped nor out of the box
  uint32_t e0, e1;                 // variant 13: the entry cell's table entry (layout: head of this file), count 0 if the ray misses the box
  float t_box;                     // variant 13: the ray parameter at which it leaves the grid box (+ 2 slacks)
  bool forced;                     // variant 13: the walk was cut short by the safety cap on its rounds -- the result is not to be trusted
  __device__ __forceinline__ bool busy() const { return (k0 < k1) | have_next | walking; }
};

// `prim` (variant 13, round 5): the ray is a PRIMARY ray of a pixel whose list of visible grid spheres was built before the sample
// loop (pt_primlist.h).  Such a lane does not walk: the "cell" it starts in is its list (same entry format, links included) and
// it is marked as having left the grid, so the pooled loop tests the list's spheres and nothing else.
// `last` (variant 13, round 5): the ray is a path's last bounce -- it walks only if it may end on an emitting sphere (below).
template <bool POOLED = false>
__device__ __forceinline__ void grid_begin(GridWalk& w, const GridLds& G, F3 o, F3 d, float a, bool prim = false, bool last = false) {
  const float INF = __builtin_inff();
  const float two_a = 2.0f * a, a4 = 4.0f * a;
  const float Tlim = 1000000.0f * two_a;
  const float Tlim_hi = Tlim * 1.0000153f;
  w.a = a;
  w.s = Near2{INF, INF, 0};
  // spheres outside the grid (walls, very large or very small ones): every lane tests all of them
  {
    const int nb = (int)(G.h.n_big & 0xFFFFu);  // wave-uniform
    int k = 0;
    for (; k + 2 <= nb; k += 2) {  // in pairs: two independent dependency chains per trip
      const int i = (int)G.big[k], j = (int)G.big[k + 1];
      const float4 gi = G.bigg[k], gj = G.bigg[k + 1];  // (their own contiguous copy: the geometry does not wait for the index)
      const bool di = near2_test<true>(w.s, gi, i, o, d, a4, Tlim_hi);
      const bool dj = near2_test<true>(w.s, gj, j, o, d, a4, Tlim_hi);
      if (__builtin_expect(di | dj, 0)) near2_resolve(w.s, (di ? 1u : 0u) | (dj ? 2u : 0u), gi, i, gj, j, gj, j, o, d, a, Tlim_hi);
    }
    if (k < nb) {
      const int i = (int)G.big[k];
      const float4 gi = G.bigg[k];
      // (a wave-uniform branch around the doubted lanes' region: the join of a DIVERGENT one here is where this compiler's
      // register allocator put the copies of a live-range split in front of the exec restore -- EXACTNESS.md A.12,
      // tools/isa_exec_lint.py, which the build runs over every kernel)
      const bool doubt = near2_test<true>(w.s, gi, i, o, d, a4, Tlim_hi);
      if (__builtin_expect(__builtin_amdgcn_ballot_w64(doubt) != 0ull, 0)) {
        if (doubt) near2_exact(w.s, gi, i, o, d, a, Tlim_hi);
      }
    }
  }
  bool settled = false;
#if PT_V13_LAST_SHORTCUT
  // THE LAST BOUNCE OF A PATH NEEDS THE WALK ONLY IF IT MAY END ON AN EMITTING SPHERE (round 5; EXACTNESS.md A.16).  All that
  // outlives such a bounce is `hit` (the colour-variance update of :200 against the bare `return` of :157-161) and `color +=
  // mask * emission` of the sphere hit (:174) -- and all but a handful of a scene's spheres emit +-0, for which that sum is
  // `color` itself whichever of them is the nearest.  So: the spheres outside the grid have just been ranked; the emitting
  // spheres inside it (GridHeader::n_emis <= 32 of them: BASELINE's 1000-sphere scene has 10) are ranked on top; and if the
  // winner of THAT set
  //   * is accepted by the reference for certain (grid_end's confirmations for `last`: a hit exists),
  //   * does not emit, and
  //   * leads every other ranked sphere -- every emitting one among them -- by more than the ranking's ambiguity margin,
  // then the reference's nearest sphere is the winner or an unranked one, i.e. one that emits +-0 either way: same `hit`, same
  // colour, same two draws -- and no walk.  Anything else (no sphere outside the grid hit: open scenes; an emitting winner; a
  // near tie) walks as before.  Sets of up to 32: beyond, and in scenes without spheres outside the grid, nothing changes.
  if constexpr (POOLED) {
    if (G.h.n_emis != 0xFFFFu && (G.h.n_big & 0xFFFFu) != 0u && __builtin_amdgcn_ballot_w64(last) != 0ull) {
      if (last) {
        const int ne = (int)G.h.n_emis;  // wave-uniform
        for (int k = 0; k < ne; k++) {
          const int i = (int)G.em_list[k];
          const float4 gi = G.geom[i];
          if (near2_test<true>(w.s, gi, i, o, d, a4, Tlim_hi)) near2_exact(w.s, gi, i, o, d, a, Tlim_hi);  // (a doubted test: the reference's own)
        }
        const uint32_t emits = (G.em_bits[(uint32_t)w.s.i1 >> 5] >> ((uint32_t)w.s.i1 & 31u)) & 1u;
        settled = (w.s.T1 < INF) & (emits == 0u) & (w.s.T2 > w.s.T1 * 1.0000038f) & (w.s.T1 < Tlim * 0.99998f);
      }
    }
  }
#endif
  bool active = false;
  int cidx = 0;
  float t_out = INF;
  w.tmax0 = INF, w.tmax1 = INF, w.tmax2 = INF, w.tdel0 = INF, w.tdel1 = INF, w.tdel2 = INF;
  w.cs0 = 0, w.cs1 = 0, w.cs2 = 0;
  uint32_t left = 0x20080200u;
  const float slack_t = G.h.slack * __builtin_amdgcn_rsqf(a);
  {
  // clip against the grid box
  const float gmin[3] = {G.h.ox, G.h.oy, G.h.oz};
  const float dims_f[3] = {(float)G.h.nx, (float)G.h.ny, (float)G.h.nz};
  const int dims[3] = {(int)G.h.nx, (int)G.h.ny, (int)G.h.nz};
  const float oo[3] = {o.x, o.y, o.z}, dd[3] = {d.x, d.y, d.z};
  const float tiny = __builtin_amdgcn_sqrtf(a) * 9.094947e-13f;  // 2^-40 |d|
  float inv[3], t_in = 0.0f;
  bool par[3];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    par[k] = !(fabsf(dd[k]) > tiny);
    inv[k] = par[k] ? 0.0f : __builtin_amdgcn_rcpf(dd[k]);
    const float gmax = gmin[k] + dims_f[k] * G.h.cs;
    const float l = (gmin[k] - oo[k]) * inv[k], h = (gmax - oo[k]) * inv[k];
    const bool inside = (oo[k] >= gmin[k]) & (oo[k] <= gmax);
    const float tn = par[k] ? (inside ? -INF : INF) : fminf(l, h);
    const float tf = par[k] ? (inside ? INF : -INF) : fmaxf(l, h);
    t_in = fmaxf(t_in, tn);
    t_out = fminf(t_out, tf);
  }
  active = (t_in <= t_out + slack_t) & (t_in * two_a < Tlim_hi);
  // entry cell and DDA state
  const int stride[3] = {1, dims[0], dims[0] * dims[1]};
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const float p = oo[k] + dd[k] * t_in;
    int ci = (int)floorf((p - gmin[k]) * G.h.inv_cs);
    ci = ci < 0 ? 0 : (ci >= dims[k] ? dims[k] - 1 : ci);
    const bool fwd = dd[k] > 0.0f;
    const float bnd = gmin[k] + (float)(ci + (fwd ? 1 : 0)) * G.h.cs;
    const float tm = par[k] ? INF : (bnd - oo[k]) * inv[k];   // a parallel axis is never the exit face
    const float td = par[k] ? INF : G.h.cs * fabsf(inv[k]);
    if (k == 0) { w.tmax0 = tm; w.tdel0 = td; }
    if (k == 1) { w.tmax1 = tm; w.tdel1 = td; }
    if (k == 2) { w.tmax2 = tm; w.tdel2 = td; }
    cidx += ci * stride[k];
    const int sk = fwd ? stride[k] : -stride[k];
    if (k == 0) w.cs0 = sk;
    if (k == 1) w.cs1 = sk;
    if (k == 2) w.cs2 = sk;
    left |= (uint32_t)(fwd ? dims[k] - 1 - ci : ci) << (10 * k);
  }
  cidx = active ? cidx : 0;
  }
  w.cidx = cidx;
  w.left = left;
  w.e0 = 0u, w.e1 = 0u;
  w.k0 = 0u, w.k1 = 0u;
  if constexpr (POOLED) {
    const uint2 e = G.cells[prim ? G.h.prim_base + (uint32_t)kPrimEntriesPerLane * threadIdx.x : (uint32_t)cidx];
    active = active & !settled;
    w.e0 = (active | prim) ? e.x : 0u;
    w.e1 = e.y;
    active = active & !prim;
  } else {
    w.k0 = G.cell_start[cidx], w.k1 = G.cell_start[cidx + 1];
    if (!active) w.k1 = w.k0;
  }
  w.n0 = 0, w.n1 = 0;
  w.have_next = false;
#ifdef PT_TIMING_ONLY_NO_WALK   // never defined in a shipped build: what the kernel costs without the grid walk
  active = false;
  w.k1 = w.k0;
  w.e0 = 0u;
#endif
  w.walking = active;
  w.t_box = t_out + 2.0f * slack_t;
  w.forced = false;
  PT_STAT(0, 1);
  PT_STAT(5, __builtin_popcountll(__builtin_amdgcn_ballot_w64(active)));
}

// The loop.  in_walk: the lane has a walk to advance (variant 11: every lane that called).  PARK (variant 12): leave as soon
// as `park` or more of the wave's lanes have nothing left to do -- their owner re-arms them with the next ray and comes back;
// PARK = false runs until no lane has anything left.
template <bool PARK>
__device__ __forceinline__ void grid_trips(GridWalk& walk, const GridLds& G, F3 o, F3 d, bool in_walk, bool alive, int park) {
  const float two_a = 2.0f * walk.a, a4 = 4.0f * walk.a;
  const float Tlim_hi = 1000000.0f * two_a * 1.0000153f;
  const float slack_t = G.h.slack * __builtin_amdgcn_rsqf(walk.a);
  // Every field in a local of its own, written back at the end: read through the reference, the step's
  // `a0 ? cs0 : (a1 ? cs1 : cs2)` becomes a select of ADDRESSES followed by a load from scratch memory (the struct is
  // scalarised only after inlining, the select is rewritten before) -- hundreds of cycles in the innermost loop.
  Near2 s = walk.s;
  float tmax0 = walk.tmax0, tmax1 = walk.tmax1, tmax2 = walk.tmax2;
  const float tdel0 = walk.tdel0, tdel1 = walk.tdel1, tdel2 = walk.tdel2;
  int cidx = walk.cidx;
  const int cs0 = walk.cs0, cs1 = walk.cs1, cs2 = walk.cs2;
  uint32_t left = walk.left, k0 = walk.k0, k1 = walk.k1, n0 = walk.n0, n1 = walk.n1;
  bool have_next = walk.have_next, walking = walk.walking;
  PT_HIST_DECL;
  for (;;) {
    // Stepping is done one cell AHEAD and for the whole wave at once: a round is triggered only when some lane has used
    // up its list and has no next one in hand, and in that round EVERY lane without a next list takes its step.  (One
    // round per lane and cell as they came due cost 55 rounds per walk with 5 of 64 lanes stepping in each.)  The stop rule
    // is evaluated when the step is taken, i.e. possibly before the current cell's tests have lowered T1: at worst one
    // cell more is visited than strictly needed -- more tests, same result.
    for (;;) {
      const bool swap = (k0 >= k1) & have_next;
      k0 = swap ? n0 : k0;
      k1 = swap ? n1 : k1;
      have_next = have_next & !swap;
      const bool starving = in_walk & walking & (k0 >= k1);  // have_next is false here for such a lane
      if (__builtin_amdgcn_ballot_w64(starving) == 0) break;
      const bool step = in_walk & walking & !have_next;
      PT_STATW(3, 1);
      PT_STATW(4, __builtin_popcountll(__builtin_amdgcn_ballot_w64(step)));
      PT_HIST_LANE(hist_rounds, 1);
      if (step) {
        PT_HIST_LANE(hist_steps, 1);
        const float t_exit = fminf(fminf(tmax0, tmax1), tmax2);
        const float reach = (t_exit - slack_t) * two_a;
        // nothing that could still matter lies beyond the cell being left: stop (same rule as before)
        const bool stop = (s.T1 * 1.0000077f < reach) | (reach > Tlim_hi);
        const bool a0 = (tmax0 <= tmax1) & (tmax0 <= tmax2);
        const bool a1 = !a0 & (tmax1 <= tmax2);
        tmax0 = a0 ? tmax0 + tdel0 : tmax0;
        tmax1 = a1 ? tmax1 + tdel1 : tmax1;
        tmax2 = (a0 | a1) ? tmax2 : tmax2 + tdel2;
        cidx += a0 ? cs0 : (a1 ? cs1 : cs2);
        left -= a0 ? 1u : (a1 ? (1u << 10) : (1u << 20));
        walking = !stop & ((left & 0x20080200u) == 0x20080200u);
        if (walking) {
          n0 = G.cell_start[cidx];
          n1 = G.cell_start[cidx + 1];
          have_next = true;
        }
      }
    }
    const bool testing = in_walk & (k0 < k1);
    const uint64_t testing_mask = __builtin_amdgcn_ballot_w64(testing);
    if (testing_mask == 0) break;  // no lane has a list left, and none can get one
    if constexpr (PARK) {
      // lanes of unfinished pixels with nothing left to do here, against the parking threshold
      if (__builtin_popcountll(__builtin_amdgcn_ballot_w64(alive & !(in_walk & ((k0 < k1) | have_next | walking)))) >= park) break;
    }
    PT_STATW(1, 1);
    PT_STATW(2, __builtin_popcountll(testing_mask));
    PT_HIST_LANE(hist_trips, 1);
    if (testing) {
      PT_HIST_LANE(hist_tests, (int)((k1 - k0) < 3u ? (k1 - k0) : 3u));
      // (Measured and dropped: requesting the NEXT trip's two indices before the tests, to take one LDS latency off the
      // chain -- 3 % slower: the extra selects and the stale-list check cost more than the latency six waves already hide.)
      // Up to three registered spheres per trip, as many as the list has left: index and geometry reads of all of them are
      // issued first, so the tests' dependency chains overlap (the loop is latency-bound).  1 -> 2 per trip: -9 %, 2 -> 3: -3 %.
      const bool two = k0 + 1u < k1;
#if PT_GRID_TESTS_PER_TRIP >= 3
      const bool three = k0 + 2u < k1;
      const int i = (int)G.items[k0];
      const int j = (int)G.items[two ? k0 + 1u : k0];
      const int l = (int)G.items[three ? k0 + 2u : k0];
      const float4 gi = G.geom[i], gj = G.geom[j], gl = G.geom[l];
      k0 += three ? 3u : (two ? 2u : 1u);
      // (the three flags stay lane masks in scalar registers; the bit field is formed only inside the rare block)
      const bool di = near2_test(s, gi, i, o, d, a4, Tlim_hi);
      bool dj = false, dl = false;
      if (two) dj = near2_test(s, gj, j, o, d, a4, Tlim_hi);
      if (three) dl = near2_test(s, gl, l, o, d, a4, Tlim_hi);
      if (__builtin_expect(di | dj | dl, 0))
        near2_resolve(s, (di ? 1u : 0u) | (dj ? 2u : 0u) | (dl ? 4u : 0u), gi, i, gj, j, gl, l, o, d, walk.a, Tlim_hi);
#else
      const int i = (int)G.items[k0];
      const int j = (int)G.items[two ? k0 + 1u : k0];
      const float4 gi = G.geom[i], gj = G.geom[j];
      k0 += two ? 2u : 1u;
      const bool di = near2_test(s, gi, i, o, d, a4, Tlim_hi);
      bool dj = false;
      if (two) dj = near2_test(s, gj, j, o, d, a4, Tlim_hi);
      if (__builtin_expect(di | dj, 0)) near2_resolve(s, (di ? 1u : 0u) | (dj ? 2u : 0u), gi, i, gj, j, gj, j, o, d, walk.a, Tlim_hi);
#endif
    }
  }
  PT_HIST_END(in_walk);
  walk.s = s;
  walk.tmax0 = tmax0, walk.tmax1 = tmax1, walk.tmax2 = tmax2;
  walk.cidx = cidx;
  walk.left = left, walk.k0 = k0, walk.k1 = k1, walk.n0 = n0, walk.n1 = n1;
  walk.have_next = have_next, walk.walking = walking;
}

__device__ __forceinline__ bool grid_end(const GridWalk& w, const SceneLds& sc, const GridLds& G, int n, F3 o, F3 d,
                                         float& t_hit, int& idx) {
  const float INF = __builtin_inff();
  const float Tlim = 1000000.0f * (2.0f * w.a);
  const Near2& s = w.s;
  const bool has = s.T1 < INF;
  bool ambiguous = (has & ((s.T2 <= s.T1 * 1.0000038f) | (s.T1 >= Tlim * 0.99998f))) | w.forced;  // (forced: variant 13's round cap)
  float t = 0.0f;
  const float4 gw = G.geom[s.i1];
  bool hit = has;
  idx = s.i1;
  {
    bool bad = false;
    const RayConst rc = make_ray_const(d);
    bool real = intersect_sphere_nb(o, d, rc, gw, t, bad);
    // the cheap sequences met an input outside their verified domain: the literal test, for the winner alone
    if (__builtin_expect(has & bad, 0)) real = intersect_sphere(o, d, rc.a, gw, t);
    const bool good = real & (t >= kMinGoodT) & (t < 1000000.0f);  // (quotient_to_float_nb relies on this range test)
    ambiguous = ambiguous | (has & !good);
    hit = has & good;
  }
  t_hit = t;
  PT_STAT(6, __builtin_popcountll(__builtin_amdgcn_ballot_w64(ambiguous)));
#ifdef PT_GRID_STATS_AMBIG  // why lanes are ambiguous (tools/grid_stats.py ambig): slots 1-4 and 7 re-used
  PT_STAT(2, __builtin_popcountll(__builtin_amdgcn_ballot_w64(has & (s.T2 <= s.T1 * 1.0000038f))));
  PT_STAT(3, __builtin_popcountll(__builtin_amdgcn_ballot_w64(has & (s.T1 >= Tlim * 0.99998f))));
  PT_STAT(4, __builtin_popcountll(__builtin_amdgcn_ballot_w64(has & bad)));
  PT_STAT(7, __builtin_popcountll(__builtin_amdgcn_ballot_w64(has & !good)));
#endif
#ifndef PT_TIMING_ONLY_NO_AMBIG  // never defined in a shipped build: what the literal redo of ambiguous lanes costs
  if (__builtin_expect(ambiguous, 0)) hit = intersect_scene_loop<0>(sc, n, o, d, make_ray_const(d), t_hit, idx);
#endif
  return hit;
}

__device__ __forceinline__ bool intersect_scene_grid(const SceneLds& sc, const GridLds& G, int n, F3 o, F3 d, float a,
                                                     float& t_hit, int& idx) {
  GridWalk w;
  grid_begin(w, G, o, d, a);
  grid_trips<false>(w, G, o, d, true, true, 0);
  return grid_end(w, sc, G, n, o, d, t_hit, idx);
}

// is this ray one the grid may be used for?  (finite, and starting within the admission radius the registration margins assume)
__device__ __forceinline__ bool grid_admits(const GridLds& G, F3 o, F3 d, float a) {
  const F3 oc = mk3(o.x - G.h.cx, o.y - G.h.cy, o.z - G.h.cz);
  const float INF = __builtin_inff();
  return (dot(oc, oc) <= G.h.far2) & (a > 0.0f) & (a < 1e30f) & (fabsf(d.x) < INF) & (fabsf(d.y) < INF) & (fabsf(d.z) < INF);
}

// variant 11's nearest-hit search: the grid when the build produced one, the brute-force loop otherwise
__device__ __forceinline__ bool intersect_scene_v11(const SceneLds& sc, int n, F3 o, F3 d, float& t_hit, int& idx) {
  if (n <= 0) return false;
  const float a = dot(d, d);
  const GridLds& G = *sc.grid;
  if (G.valid) {
    const bool admitted = grid_admits(G, o, d, a);
    PT_STATW(7, __builtin_amdgcn_ballot_w64(!admitted) != 0 ? 1 : 0);  // waves that also run the brute-force loop
    if (admitted) return intersect_scene_grid(sc, G, n, o, d, a, t_hit, idx);
  }
  return intersect_scene_screened_large(sc, n, o, d, make_ray_const(d), t_hit, idx);
}

// ---- variant 13: the grid walk with the sphere tests POOLED across the lanes of the wave ---------------------------------
// Variant 11's walk is divergent: a ray needs 12 sphere tests on average but one ray in a hundred needs 55, so a wave makes
// 21 test trips (up to three tests each) with 17 of its 64 lanes testing -- and the tests are two thirds of the kernel's
// instructions (tools/grid_stats.py, profiles/r03).  Here a lane does not test its own spheres.  Rays, accumulators and
// generators stay in their lanes (a pixel's float sums are order-sensitive); what moves is the WORK:
//  * a lane that steps into a cell appends that cell's registered spheres to a per-wave LDS ring as (owner lane, sphere)
//    entries -- slots from a ballot and v_mbcnt prefix per list position, no atomics, no scan;
//  * whenever the ring holds a wave's worth, all 64 lanes drain it: lane r takes entry head + r, fetches the OWNER's ray with
//    ds_bpermute (the wave64 __shfl), runs the same float screen near2_test runs, and returns a hit to the owner's slot
//    with LDS atomics on integer keys: key = (estimate's float bits << 32) | sphere index, best = ds_min_u64, and the
//    loser of each exchange goes to the runner-up slot with ds_min_u32.  Best and runner-up are the two smallest elements
//    of the MULTISET of tested keys, so they do not depend on the order in which the wave happened to test them; equal keys
//    are the same sphere met in two cells and are entered once (what near2_test's leader check does);
//  * the owner reads its best estimate back after a drain, for the stop rule of its next steps.  That value may be stale by
//    the entries still in the ring: the rule then fires later, never earlier -- more cells, more tests, the same superset
//    argument as the look-ahead step of variant 11 (EXACTNESS.md A.6 (iii-b)).
// Round 4.  Counters (profiles/r04): the walk is ISSUE-bound, not latency-bound -- 71 vector instructions per round and cell
// (24.6 rounds per wave-walk with 16.5 of 64 lanes stepping), most of them at the half rate (compares, selects), against 48 per
// drain of 60 tests.  So the rounds were rebuilt for instruction count:
//  * a cell's table entry carries its registrations inline (head of this file): one ds_read_b64 per visit instead of two index
//    reads and up to three list reads with their address arithmetic, and the push takes its ring values from registers;
//  * cells with more than three registrations (3 %) continue in chained entries: a lane that holds a link reads that entry
//    instead of taking a DDA step -- no list ranges, no second code path;
//  * a lane advances PT_POOL_STEPS = 2 cells per round: both steps are pure arithmetic on the DDA state, both entries are
//    requested together and one wait, one loop trip and one round of bookkeeping serve them.  The stop rule of the second step
//    sees the best estimate the first saw: at worst one cell more than strictly needed -- more tests, same result (A.6 (iii-b));
//  * ring positions come from three compares and six chained v_mbcnt, the three stores of a push are unconditional (a lane that
//    has fewer than three spheres writes garbage into slots that a later store of the same push overwrites: stores go out in
//    descending slot order, and a slot's rightful owner always holds a lower slot number than the garbage that hits it);
//  * after a drain the ring's remainder (less than a pass) moves to the ring's start, so positions need no wrap-around;
//  * the best estimate is read back only after a drain (nothing else can change it).
// (The walk-range hand-over of round 3's lab build -- idle lanes taking the far half of a busy lane's walk -- is gone: a
// measured negative result, HISTORY.md B.5.)
// Doubted tests (origin within rounding distance of a surface) are decided on the spot by the reference's own FP64
// expression, by the lane that drew the entry, from the owner's ray: same operands, same bits.  Everything after the walk
// (ambiguity rule, exact step on the winner, literal fallback) is grid_end, unchanged.
#ifndef PT_POOL_STEPS
#define PT_POOL_STEPS 2
#endif
#ifdef PT_DEBUG_PIXEL  // never defined in a shipped build: device printf for one pixel's walks (block, thread given on the command line)
#define PT_DBG_ME() (blockIdx.x == PT_DEBUG_BLOCK && threadIdx.x == PT_DEBUG_THREAD)
#define PT_DBG_MY_WAVE() (blockIdx.x == PT_DEBUG_BLOCK && (threadIdx.x >> 6) == (PT_DEBUG_THREAD >> 6))
#define PT_DBG_OWNED(owner) (PT_DBG_MY_WAVE() && (owner) == (PT_DEBUG_THREAD & 63))
#define PT_DBG(...) do { printf(__VA_ARGS__); } while (0)
#else
#define PT_DBG_ME() false
#define PT_DBG_MY_WAVE() false
#define PT_DBG_OWNED(owner) false
#define PT_DBG(...) do { } while (0)
#endif
#ifndef PT_POOL_SCAN_DPP
#define PT_POOL_SCAN_DPP 1  // the sweep's prefix sum: 1 = DPP row shifts and broadcasts, 0 = six ds_bpermute steps
#endif
#ifndef PT_POOL_SWEEP_BELOW
#define PT_POOL_SWEEP_BELOW 24  // the sweep (grid_trips_pooled, (2b)) takes over once at most this many lanes of the wave still walk
#endif
static_assert(63 + 64 * 3 * PT_POOL_STEPS + 2 <= kPoolRing, "pool ring too small");
static_assert(PT_POOL_STEPS == 1 || PT_POOL_STEPS == 2, "the link queue has two slots");

struct PoolLds {
  uint32_t* ring;            // [kPoolRing] (owner lane << 16) | sphere index
  unsigned long long* key1;  // [64] per owner lane: (bits of the smallest estimate << 32) | its sphere
  uint32_t* t2;              // [64] per owner lane: bits of the second smallest estimate
  uint32_t* olist;           // [64] the sweep: per ray of a pass, owner lane | first slot << 8 | crossings already served << 16
  unsigned long long* smask; // the sweep: bit s set = a ray's crossings start at helper slot s
};
constexpr unsigned long long kPoolEmpty = 0x7F800000FFFFFFFFull;  // +inf, no sphere

__device__ __forceinline__ PoolLds pool_of_wave(void* workgroup_base) {
  char* b = reinterpret_cast<char*>(workgroup_base) + (threadIdx.x >> 6) * kPoolWaveBytes;
  PoolLds p;
  p.key1 = reinterpret_cast<unsigned long long*>(b);
  p.ring = reinterpret_cast<uint32_t*>(b + 64 * 8);
  p.t2 = reinterpret_cast<uint32_t*>(b + 64 * 8 + kPoolRing * 4);
  p.olist = p.t2 + 64;
  p.smask = reinterpret_cast<unsigned long long*>(p.olist + 64);
  return p;
}

__device__ __forceinline__ float bperm_f(int byte_addr, float v) {
  return __int_as_float(__builtin_amdgcn_ds_bpermute(byte_addr, __float_as_int(v)));
}

// One DDA step of the pooled walk; returns whether the lane still walks, i.e. stands in a new cell.  Round 4: the whole stop
// rule is ONE compare of the ray parameter at which the lane leaves its cell against `tstop`, the smallest of
//   * the parameter beyond which nothing can beat or tie the best estimate: T1 (1 + 2^-17) / 2a + slack_t  (grid_trips' rule
//     "T1 (1 + 2^-17) < 2a (t_exit - slack_t)" solved for t_exit),
//   * the 1e6 acceptance limit in the same units,
//   * the parameter at which the ray leaves the grid box, plus two slacks (instead of per-axis cell counters: the box is the
//     spheres' bounding box inflated by more than a slack beyond every registration, so nothing is registered out there),
// all three rounded UP (a later stop visits more cells: more tests, same result).  A lane may thus step one cell past the box
// before the test fires; the cell index it READS is clamped into the table (any cell's spheres are valid extra tests).
// Termination: tmax grows by tdel >= cs / |d| per step of its axis, a normal float for every admitted ray (|d| < 1e15), and a
// NaN fails the compare -- and the caller caps the rounds of a walk regardless (GridWalk::forced).
// (A free function with the strides BY VALUE: inside a lambda that captures them by reference the select between them becomes a
// select of addresses and a load from scratch memory, see grid_trips.)
__device__ __forceinline__ bool dda_step(float& tmax0, float& tmax1, float& tmax2, float tdel0, float tdel1, float tdel2, int& cidx,
                                         int cs0, int cs1, int cs2, float tstop) {
  const float t_exit = fminf(fminf(tmax0, tmax1), tmax2);
  const bool a0 = (tmax0 <= tmax1) & (tmax0 <= tmax2);
  const bool a1 = !a0 & (tmax1 <= tmax2);
  tmax0 = a0 ? tmax0 + tdel0 : tmax0;
  tmax1 = a1 ? tmax1 + tdel1 : tmax1;
  tmax2 = (a0 | a1) ? tmax2 : tmax2 + tdel2;
  cidx += a0 ? cs0 : (a1 ? cs1 : cs2);
  return t_exit <= tstop;
}

// The walk of variant 13 (between grid_begin<true> and grid_end).  Every lane that is in the function helps testing; `walk`
// belongs to the lane's own ray (walk.walking false and an empty entry: a ray that misses the grid box).
__device__ __forceinline__ void grid_trips_pooled(GridWalk& walk, const GridLds& G, const PoolLds& P, F3 o, F3 d) {
  constexpr int K = PT_POOL_STEPS;
  const float INF = __builtin_inff();
  const int lane = threadIdx.x & 63;
  const float a4 = 4.0f * walk.a;  // what the testers fetch
  const float two_a = 2.0f * walk.a;
  const float slack_t = G.h.slack * __builtin_amdgcn_rsqf(walk.a);
  const float Tcap = 1000000.0f * two_a * 1.0000153f;  // the walk ends where 2a t passes the 1e6 limit
  const float inv_two_a = __builtin_amdgcn_rcpf(two_a);
  float tmax0 = walk.tmax0, tmax1 = walk.tmax1, tmax2 = walk.tmax2;
  const float tdel0 = walk.tdel0, tdel1 = walk.tdel1, tdel2 = walk.tdel2;
  int cidx = walk.cidx;
  const int cs0 = walk.cs0, cs1 = walk.cs1, cs2 = walk.cs2;
  const uint32_t last_cell = G.h.nx * G.h.ny * G.h.nz - 1u;
  const float t_box = walk.t_box;
  // where the walk stops (dda_step), from the best estimate as of the last drain; 1 + 2^-19 covers the roundings of this line
  auto stop_at = [&](float best) { return fminf(fmaf(fminf(best * 1.0000077f, Tcap), inv_two_a, slack_t), t_box) * 1.0000019f; };
  bool walking = walk.walking;
  // safety net: a monotone walk crosses at most nx + ny + nz cells; a wave that needs more rounds than that holds a lane whose
  // DDA does not advance (no admitted ray does that) -- such lanes are stopped and their result is left to the literal loop
  // (plus the chained table entries: a lane that follows a link does not step in that round)
  int rounds_left = (int)(G.h.nx + G.h.ny + G.h.nz) / K + 8 + (int)((G.h.n_big >> 16) - (last_cell + 1u));
  // lanes of this wave that are here (the others are on the brute-force path, or their pixel is finished): ranks, not lane
  // numbers, index the ring
  const uint64_t here = __builtin_amdgcn_ballot_w64(true);
  const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(here >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)here, 0u));
  const uint32_t n_here = (uint32_t)__builtin_popcountll(here);
  // what the spheres outside the grid (grid_begin) left: the owner's slots
  // (every access to the slots is an atomic one: they are written by whichever lanes draw the owner's entries)
  __hip_atomic_store(P.key1 + lane, walk.s.T1 < INF ? (((unsigned long long)__float_as_uint(walk.s.T1) << 32) | (uint32_t)walk.s.i1) : kPoolEmpty,
                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  __hip_atomic_store(P.t2 + lane, __float_as_uint(walk.s.T2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  uint32_t links = 0u;   // chained table entries this lane still has to follow: a queue of two 16-bit slots (see (3))
  uint32_t tail = 0u;    // wave-uniform: entries in the ring, which always starts at slot 0 when a push begins
  float tstop = stop_at(walk.s.T1);  // (the owner's best estimate as of the last drain: nothing but a drain changes it)
  PT_HIST_DECL;
  if (PT_DBG_ME()) PT_DBG("ENTRY walking %d tmax %g %g %g tdel %g %g %g cidx %d cs %d %d %d tstop %g t_box %g e0 %08x e1 %08x n_here %u T1 %g o %g %g %g d %g %g %g\n", (int)walking, tmax0, tmax1, tmax2, tdel0, tdel1, tdel2, cidx, cs0, cs1, cs2, tstop, t_box, walk.e0, walk.e1, n_here, walk.s.T1, o.x, o.y, o.z, d.x, d.y, d.z);
  // (1) append an entry's spheres to the ring.  Slots: a lane's are consecutive and start at the exclusive prefix sum of the
  // counts, which three compares and six chained v_mbcnt give (count >= 1, >= 2, >= 3: the count sits in the top two bits).
  // `tag`: the owner of the ray the entry was read for, << 16.  `first`: which of the two link slots a link goes to.
  auto push = [&](uint32_t e0k, uint32_t e1k, uint32_t tag, bool first) {
    const uint64_t m1 = __builtin_amdgcn_ballot_w64(e0k >= 0x40000000u);
    if (m1 != 0) {
      const uint64_t m2 = __builtin_amdgcn_ballot_w64(e0k >= 0x80000000u), m3 = __builtin_amdgcn_ballot_w64(e0k >= 0xC0000000u);
      uint32_t pos = __builtin_amdgcn_mbcnt_hi((uint32_t)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m1, tail));
      pos = __builtin_amdgcn_mbcnt_hi((uint32_t)(m2 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m2, pos));
      pos = __builtin_amdgcn_mbcnt_hi((uint32_t)(m3 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m3, pos));
      PT_HIST_LANE(hist_tests, (int)(e0k >> 30));
      if (e0k >= 0x40000000u) {
        // (descending, and THREE instructions: see the head of this section.  The barriers keep the compiler from fusing two of
        // the stores into one ds_write2_b32, inside which a lane's garbage and its neighbour's rightful value would race.)
        uint32_t* slot = P.ring + pos;
        slot[2] = tag | (e1k >> 16);
        asm volatile("" ::: "memory");
        slot[1] = tag | (e1k & 0xFFFFu);
        asm volatile("" ::: "memory");
        slot[0] = tag | (e0k & 0xFFFFu);
      }
      tail += (uint32_t)(__builtin_popcountll(m1) + __builtin_popcountll(m2) + __builtin_popcountll(m3));
      // a link goes to the queue slot of its entry: entry 0's to the high half (where the entry word already has it), entry 1's
      // to the low half.  (Only an entry with spheres has a link; the slot is free: see (3).)
      if (first) links |= e0k & 0x1FFF0000u; else links |= (e0k >> 16) & 0x1FFFu;
    }
  };
  // (2) drain: whole passes while the ring holds one; everything once nobody can add to it any more (fin)
  auto drain = [&](bool fin) {
  if (tail >= (fin ? 1u : n_here)) {
    uint32_t head = 0u;
    do {
      const uint32_t n = (tail - head) < n_here ? (tail - head) : n_here;
      const bool valid = rank < n;
      PT_STATW(1, 1);
      PT_STATW(2, n);
      PT_HIST_LANE(hist_trips, 1);
      const uint32_t e = P.ring[head + rank];
      head += n;
      const int owner = valid ? (int)(e >> 16) : lane;
      const int i = valid ? (int)(e & 0xFFFFu) : 0;
      const float4 g = G.geom[i];
      const int oa = owner << 2;
      const F3 ro = mk3(bperm_f(oa, o.x), bperm_f(oa, o.y), bperm_f(oa, o.z));
      const F3 rd = mk3(bperm_f(oa, d.x), bperm_f(oa, d.y), bperm_f(oa, d.z));
      const float ra4 = bperm_f(oa, a4);
      const float rTlim_hi = 1000000.0f * (0.5f * ra4) * 1.0000153f;  // the owner's own Tlim_hi: same operands, same operations
      // the float screen of near2_test
      const F3 off = mk3(ro.x - g.x, ro.y - g.y, ro.z - g.z);
      const float b = 2.0f * dot(rd, off);
      const float cc = dot(off, off) - g.w;
      const float bb = b * b;
      const float a4c = ra4 * cc;
      const float dacc = fmaf(-ra4, cc, bb);
      const bool cand = valid & ((int)__float_as_uint(dacc) >= 0);
      const float sq = __builtin_amdgcn_sqrtf(dacc);
      const float ee = fmaf(b, b, -bb);
      const float num = a4c + ee;
      const float TA = copysign_neg_b3(sq, b) - b;  // -q (near2_test)
      const float TB = num * __builtin_amdgcn_rcpf(TA);
      const uint32_t ta = __float_as_uint(TA), tb = __float_as_uint(TB);
      const uint32_t tbits = ta < tb ? ta : tb;  // the smaller positive root's bits, sign bit set or NaN bits if none (near2_test)
      float T = __uint_as_float(tbits);
      const bool sure = fabsf(a4c) > fmaf(bb, 1.1920929e-07f, 1e-30f);
      bool ok = cand & sure & (tbits < __float_as_uint(rTlim_hi));
      if (__builtin_expect(cand & !sure, 0)) {  // near2_exact: the reference's own test, 2a*t in place of the estimate
        float t = 0.0f;
        const float ra = 0.25f * ra4;
        const bool h = intersect_sphere(ro, rd, ra, g, t);
        T = (2.0f * ra) * t;
        ok = h & (t > 0.0f) & (T < rTlim_hi);
      }
      if (ok) {
        const unsigned long long key = ((unsigned long long)__float_as_uint(T) << 32) | (uint32_t)i;
        const unsigned long long old = __hip_atomic_fetch_min(P.key1 + owner, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (old != key) {  // (an equal key is the same sphere met in another cell: entered once)
          const unsigned long long loser = old > key ? old : key;
          __hip_atomic_fetch_min(P.t2 + owner, (uint32_t)(loser >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      }
    } while ((tail - head) >= (fin ? 1u : n_here));
    // what is left (less than a pass) moves to the ring's start (DS operations of one wave execute in order)
    const uint32_t rest = tail - head;
    if (rest != 0u) {
      const uint32_t v = P.ring[head + (rank < rest ? rank : 0u)];
      if (rank < rest) P.ring[rank] = v;
    }
    tail = rest;
    tstop = stop_at(__uint_as_float(__hip_atomic_load(reinterpret_cast<uint32_t*>(P.key1 + lane) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)));
  }
  };

  // entries in hand: the cell(s) the lane has just stepped into, or the chained entries it has followed
  uint32_t e0[K], e1[K];
  e0[0] = walk.e0, e1[0] = walk.e1;
#pragma unroll
  for (int k = 1; k < K; k++) e0[k] = 0u, e1[k] = 0u;
  const uint32_t self = (uint32_t)lane << 16;
  bool sweep = false;  // wave-uniform: the lock-step rounds were left for the sweep below
  for (;;) {
#pragma unroll
    for (int k = 0; k < K; k++) push(e0[k], e1[k], self, k == 0);
    const bool fin = __builtin_amdgcn_ballot_w64(walking | (links != 0u)) == 0;
    // (Measured and dropped, round 5: requesting the next cells BEFORE the drain, so that their LDS round trip overlaps the drain's --
    // the steps then see a stop parameter one drain older, walk further and test more: closed / open 71.0 / 22.5 -> 76.5 / 24.3 ms.
    // Issuing the best estimate's read-back ahead of the remainder's read, to overlap the two round trips: no difference.
    // profiles/r05/cfg4_ab.txt)
    drain(fin);
    if (fin) break;
    if (__builtin_expect(--rounds_left < 0, 0)) {  // (wave-uniform)
      walk.forced = walk.forced | walking | (links != 0u);
      walking = false;
      links = 0u;
    }
    // few lanes still walk (none holds a link, the wave is complete): the sweep takes over
    if (n_here == 64u && __builtin_amdgcn_ballot_w64(links != 0u) == 0 &&
        __builtin_popcountll(__builtin_amdgcn_ballot_w64(walking)) <= PT_POOL_SWEEP_BELOW) {
      if (PT_DBG_ME()) PT_DBG("SWITCH walking %d walkers %d tmax %g %g %g cidx %d tstop %g rounds_left %d\n", (int)walking, (int)__builtin_popcountll(__builtin_amdgcn_ballot_w64(walking)), tmax0, tmax1, tmax2, cidx, tstop, rounds_left);
      sweep = true;
      break;
    }
    // (3) the next entries.  A lane that holds links follows ONE of them (the entry it reads may add another: its round ends
    // there) -- the high slot's if there is one, so that the slot its new entry 0 may write a link to is free; a lane without
    // links takes K DDA steps.  At most two links are ever pending: two from the K cells of a stepping round, or the one left
    // over plus the one the followed entry brought.
    PT_STATW(3, 1);
    PT_STATW(4, __builtin_popcountll(__builtin_amdgcn_ballot_w64(walking | (links != 0u))));
    PT_HIST_LANE(hist_rounds, 1);
    uint32_t cur = 0u;
    if (__builtin_amdgcn_ballot_w64(links != 0u) != 0) {
      const uint32_t hi = links >> 16;
      cur = hi != 0u ? hi : links;
      links = hi != 0u ? (links & 0xFFFFu) : 0u;
    }
#pragma unroll
    for (int k = 0; k < K; k++) e0[k] = 0u, e1[k] = 0u;
    const bool chain = cur != 0u;
    bool go = chain;
    if (!chain & walking) {
      PT_HIST_LANE(hist_steps, 1);
      walking = dda_step(tmax0, tmax1, tmax2, tdel0, tdel1, tdel2, cidx, cs0, cs1, cs2, tstop);
      go = walking;
    }
    if (go) {
      const uint32_t at = (uint32_t)cidx < last_cell ? (uint32_t)cidx : last_cell;  // (a step past the box: any cell will do)
      const uint2 e = G.cells[chain ? cur : at];
      e0[0] = e.x, e1[0] = e.y;
    }
    if (PT_DBG_ME()) PT_DBG("STEP0 walking %d chain %d cidx %d entry %08x %08x tmax %g %g %g\n", (int)walking, (int)chain, cidx, e0[0], e1[0], tmax0, tmax1, tmax2);
#pragma unroll
    for (int k = 1; k < K; k++) {
      if (!chain & walking) {
        PT_HIST_LANE(hist_steps, 1);
        walking = dda_step(tmax0, tmax1, tmax2, tdel0, tdel1, tdel2, cidx, cs0, cs1, cs2, tstop);
        if (walking) {
          const uint32_t at = (uint32_t)cidx < last_cell ? (uint32_t)cidx : last_cell;
          const uint2 e = G.cells[at];
          e0[k] = e.x, e1[k] = e.y;
        }
      }
    }
    if (PT_DBG_ME()) PT_DBG("STEP1 walking %d cidx %d entry %08x %08x tmax %g %g %g\n", (int)walking, cidx, e0[K - 1], e1[K - 1], tmax0, tmax1, tmax2);
  }
  // (2b) THE SWEEP.  Lock-step DDA rounds are cheap per cell while most lanes walk, and hopeless once few do: a round costs the
  // same 91 vector instructions with 49 lanes stepping or with 5, and a wave-walk of 13.7 rounds averages 18.  Once at most
  // PT_POOL_SWEEP_BELOW lanes still walk the wave stops stepping: every walking lane counts the cell boundaries its ray crosses on
  // each axis before tstop -- the DDA would take exactly those steps, one by one -- and the wave's 64 lanes compute the cells
  // behind these crossings IN PARALLEL, a crossing per lane: slots from a prefix sum over the rays' counts, the ray's DDA state
  // fetched with ds_bpermute, the cell from
  //   t_c = tmax_k + i tdel_k,   steps taken on another axis m by then = floor((t_c - tmax_m) / tdel_m) + 1  (0 if t_c < tmax_m),
  // its table entry read and pushed with the RAY OWNER's tag.  Same cells as the DDA up to roundings of a boundary parameter,
  // i.e. up to slivers far thinner than the registration slack (A.6 (ii)); all crossings up to tstop as of the switch are
  // taken -- no early stop on later hits: more tests, never fewer.  Links are followed by the lane that drew the entry.
  // (A loop of its own, after the lock-step one: what it keeps per lane -- counts, progress, the owner tag -- costs the
  // lock-step rounds no registers, and the rounds' own state is dead here.  Measured as one loop: 7 % slower with the sweep
  // switched OFF than without its code.)
  if (sweep) {
    // n0 | n1 << 8 | n2 << 16 | (cs0 < 0) << 24 | (cs1 < 0) << 25 | (cs2 < 0) << 26, their sum, how many of them are served
    uint32_t sw_n = 0u, sw_total = 0u, sw_done = 0u;
    bool wide = false;
    if (walking) {
      auto crossings = [&](float tm, float td) {  // boundaries at tm, tm + td, ... not beyond tstop (dda_step: t_exit <= tstop)
        const int n = tm <= tstop ? (int)((tstop - tm) * __builtin_amdgcn_rcpf(td)) + 1 : 0;
        return (uint32_t)(n < 0 ? 0 : n);
      };
      const uint32_t n0 = crossings(tmax0, tdel0), n1 = crossings(tmax1, tdel1), n2 = crossings(tmax2, tdel2);
      wide = (n0 | n1 | n2) > 255u;
      sw_n = n0 | (n1 << 8) | (n2 << 16) | (cs0 < 0 ? 1u << 24 : 0u) | (cs1 < 0 ? 1u << 25 : 0u) | (cs2 < 0 ? 1u << 26 : 0u);
      sw_total = n0 + n1 + n2;
      if (PT_DBG_ME()) PT_DBG("COUNTS n %u %u %u\n", n0, n1, n2);
    }
    // (a count that does not fit its byte -- a grid of more than 255 cells along an axis, crossed end to end: the result is left
    // to the literal loop, like a walk the safety net has cut short)
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(wide) != 0, 0)) {
      walk.forced = walk.forced | walking;
      sw_total = 0u;
    }
    uint32_t se0 = 0u, se1 = 0u, tag = self;
    for (;;) {
      // the next entries: a link round (the lanes that drew entries with links read on, tag kept) or a sweep pass
      se0 = 0u, se1 = 0u;
      if (__builtin_amdgcn_ballot_w64(links != 0u) != 0) {
        if (links != 0u) {
          const uint32_t hi = links >> 16, cur = hi != 0u ? hi : links;
          links = hi != 0u ? (links & 0xFFFFu) : 0u;
          const uint2 e = G.cells[cur];
          se0 = e.x, se1 = e.y;
        }
      } else {
        // helper slot j (= lane j: the wave is complete) serves one crossing
        const uint32_t rem = sw_total - sw_done;
        uint32_t incl = rem;  // inclusive prefix sum over the lanes
#if PT_POOL_SCAN_DPP
        // in the vector ALU's data-parallel primitives: within rows of 16 by shifts of 1, 2, 4, 8, then row 0's total into row 1 and
        // row 2's into row 3, then the first half's into the second
        incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x111, 0xF, 0xF, false);  // row_shr:1
        incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x112, 0xF, 0xF, false);  // row_shr:2
        incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x114, 0xF, 0xF, false);  // row_shr:4
        incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x118, 0xF, 0xF, false);  // row_shr:8
        incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x142, 0xA, 0xF, false);  // row_bcast:15 into rows 1 and 3
        incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x143, 0xC, 0xF, false);  // row_bcast:31 into rows 2 and 3
#else
        {
          uint32_t me = (uint32_t)lane;
          asm volatile("" : "+v"(me));  // (keeps the six addresses and masks of the scan from being hoisted out of the kernel's loops and spilled)
#pragma unroll
          for (uint32_t dlt = 1; dlt < 64u; dlt <<= 1) {
            const uint32_t up = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((me - dlt) & 63u) << 2), (int)incl);
            incl += me >= dlt ? up : 0u;
          }
        }
#endif
        const uint32_t excl = incl - rem;
        const uint32_t take = excl < 64u ? (rem < 64u - excl ? rem : 64u - excl) : 0u;  // this ray's crossings served in this pass
        const uint64_t tm = __builtin_amdgcn_ballot_w64(take != 0u);
        if (lane == 0) *P.smask = 0ull;
        if (take != 0u) {
          const uint32_t r = __builtin_amdgcn_mbcnt_hi((uint32_t)(tm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)tm, 0u));
          P.olist[r] = (uint32_t)lane | (excl << 8) | (sw_done << 16);
          __hip_atomic_fetch_or(P.smask, 1ull << excl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        const uint32_t served = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);  // all remaining crossings of the wave
        const unsigned long long starts = __hip_atomic_load(P.smask, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(starts >> 1)),
                       hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(starts >> 33));
        // rays are packed in lane order and the first starts at slot 0: the ray of slot j is the (number of starts in 1..j)-th
        const uint32_t ray = __builtin_amdgcn_mbcnt_hi(hi, __builtin_amdgcn_mbcnt_lo(lo, 0u));
        const bool serving = (uint32_t)lane < (served < 64u ? served : 64u);
        PT_STATW(3, 1);  // (instrumented builds count a pass as a round, its crossings as the lanes stepping)
        PT_STATW(4, served < 64u ? served : 64u);
        sw_done += take;
        // (the fetches by EVERY lane, serving or not: ds_bpermute reads nothing from a lane that is masked off, and a ray's owner
        // need not be among the serving lanes)
        const uint32_t info = P.olist[serving ? ray : 0u];
        const int oa = (int)(info & 0xFFu) << 2;
        const float m0 = bperm_f(oa, tmax0), m1 = bperm_f(oa, tmax1), m2 = bperm_f(oa, tmax2);
        const float d0 = bperm_f(oa, tdel0), d1 = bperm_f(oa, tdel1), d2 = bperm_f(oa, tdel2);
        const int base = __builtin_amdgcn_ds_bpermute(oa, cidx);
        const uint32_t nn = (uint32_t)__builtin_amdgcn_ds_bpermute(oa, (int)sw_n);
        if (serving) {
          const uint32_t li = (info >> 16) + ((uint32_t)lane - ((info >> 8) & 0xFFu));  // which of the ray's crossings
          const uint32_t n0 = nn & 0xFFu, n1 = (nn >> 8) & 0xFFu;
          const bool k0 = li < n0, k1 = !k0 & (li < n0 + n1);
          const uint32_t i = li - (k0 ? 0u : (k1 ? n0 : n0 + n1));
          const float tc = (k0 ? m0 : (k1 ? m1 : m2)) + (float)i * (k0 ? d0 : (k1 ? d1 : d2));
          // How many crossings of another axis m come BEFORE this one: the quotient, biased DOWN by a quarter, is the count or
          // one short of it (its error is far below that: 255 crossings at most, and a boundary parameter within ulp(t) / tdel of
          // the real one); the crossing in question then settles it, its parameter formed exactly as that crossing forms its own
          // tc -- so that every pair of crossings agrees on which of the two comes first (equal parameters: the lower axis).
          // Without that, two crossings with (nearly) equal parameters can each put itself first, and the cell behind BOTH of
          // them -- where the ray goes on, for a whole cell's length -- is never visited (seen: tools/grid_check.py
          // random150_open, one pixel; tests/test_parity_gpu.py::test_sweep_crossings_with_equal_parameters).
          const int ax = k0 ? 0 : (k1 ? 1 : 2);
          auto taken = [&](float tmx, float td, int m) {
            const int j = tc >= tmx ? (int)((tc - tmx) * __builtin_amdgcn_rcpf(td) + 0.75f) : 0;
            const float tj = tmx + (float)j * td;  // crossing j of axis m (NaN on an axis the ray is parallel to: no change)
            return ((tj < tc) | ((tj == tc) & (m < ax))) ? j + 1 : j;
          };
          const int c0 = k0 ? (int)i + 1 : taken(m0, d0, 0), c1 = k1 ? (int)i + 1 : taken(m1, d1, 1),
                    c2 = (k0 | k1) ? taken(m2, d2, 2) : (int)i + 1;
          const int sx = (int)G.h.nx, sxy = (int)(G.h.nx * G.h.ny);
          const int idx = base + ((nn >> 24) & 1u ? -c0 : c0) + ((nn >> 25) & 1u ? -c1 : c1) * sx + ((nn >> 26) & 1u ? -c2 : c2) * sxy;
          const uint32_t at = (uint32_t)idx < last_cell ? (uint32_t)idx : last_cell;
          const uint2 e = G.cells[at];
          se0 = e.x, se1 = e.y;
          tag = (info & 0xFFu) << 16;
          if (PT_DBG_OWNED(info & 0xFFu)) PT_DBG("SERVE helper %d li %u axis %d i %u tc %g c %d %d %d base %d idx %d entry %08x %08x nn %08x\n", lane, li, k0 ? 0 : (k1 ? 1 : 2), i, tc, c0, c1, c2, base, idx, e.x, e.y, nn);
        }
      }
      push(se0, se1, tag, true);
      const bool fin = __builtin_amdgcn_ballot_w64((links != 0u) | (sw_done < sw_total)) == 0;
      drain(fin);
      if (fin) break;
    }
  }
  PT_HIST_END(true);
  const unsigned long long k = __hip_atomic_load(P.key1 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  walk.s.T1 = __uint_as_float((uint32_t)(k >> 32));
  walk.s.i1 = walk.s.T1 < INF ? (int)(uint32_t)k : 0;
  walk.s.T2 = __uint_as_float(__hip_atomic_load(P.t2 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
  if (PT_DBG_ME()) PT_DBG("FINAL T1 %g i1 %d T2 %g forced %d swept %d\n", walk.s.T1, walk.s.i1, walk.s.T2, (int)walk.forced, (int)sweep);
}

// `walker`: the lane has a ray of its own.  A lane without one (its pixel's samples are done: the tail of every workgroup of a
// regeneration kernel, where lanes finish at different times) is here to HELP: it draws sphere tests and sweep crossings like
// everybody else, owns nothing and gets nothing back.
__device__ __forceinline__ bool intersect_scene_grid_pooled(const SceneLds& sc, const GridLds& G, int n, F3 o, F3 d, float a,
                                                            float& t_hit, int& idx, bool walker = true, bool prim = false,
                                                            bool last = false) {
  GridWalk w;
  // (no branch around grid_begin for the helpers: they run it on whatever ray they last had -- the wave executes it anyway -- and
  // their walk is then emptied with selects.  A divergent region that ends here, in front of the register-hungry walk, is where
  // the allocator's split copies landed in front of the exec restore: EXACTNESS.md A.12.)
  grid_begin<true>(w, G, o, d, a, prim & walker, last & walker);
  const float INF = __builtin_inff();
  w.s.T1 = walker ? w.s.T1 : INF;
  w.s.T2 = walker ? w.s.T2 : INF;
  w.walking = w.walking & walker;
  w.e0 = walker ? w.e0 : 0u;
  w.e1 = walker ? w.e1 : 0u;
  grid_trips_pooled(w, G, pool_of_wave(sc.pool), o, d);
  if (!walker) return false;
  return grid_end(w, sc, G, n, o, d, t_hit, idx);
}

// variant 13's nearest-hit search
// `live` false: a lane without a ray (see intersect_scene_grid_pooled); it returns false and its outputs are not written
// `prim`: a primary ray of a pixel with a list (grid_begin)
// `last`: only hit/miss and the index are wanted (grid_end)
__device__ __forceinline__ bool intersect_scene_v13(const SceneLds& sc, int n, F3 o, F3 d, float& t_hit, int& idx, bool live = true,
                                                    bool prim = false, bool last = false) {
  if (n <= 0) return false;
  const float a = dot(d, d);
  const GridLds& G = *sc.grid;
  if (G.valid) {
    const bool admitted = live && grid_admits(G, o, d, a);
    PT_STATW(7, __builtin_amdgcn_ballot_w64(live & !admitted) != 0 ? 1 : 0);
    // (helpers go in only where somebody walks: wave-uniform)
    const bool walk_here = __builtin_amdgcn_ballot_w64(admitted) != 0;
    if (admitted | (walk_here & !live)) {
      const bool hit = intersect_scene_grid_pooled(sc, G, n, o, d, a, t_hit, idx, admitted, prim, last);
      if (admitted) return hit;
    }
  }
  if (!live) return false;
  return intersect_scene_screened_large(sc, n, o, d, make_ray_const(d), t_hit, idx);
}

}  // namespace pt
