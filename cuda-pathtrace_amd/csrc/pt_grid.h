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

#ifndef PT_GRID_MAX_CELLS
#define PT_GRID_MAX_CELLS 2048
#endif
#ifndef PT_GRID_CELLS_PER_SPHERE
#define PT_GRID_CELLS_PER_SPHERE 2.0f
#endif
constexpr int kGridMaxCells = PT_GRID_MAX_CELLS;
constexpr int kGridMaxItems = 8192;
constexpr int kGridMaxBig = 64;
constexpr int kGridMaxSpheres = PT_GRID_MAX_SPHERES;  // geometry of all spheres is staged (16 B each)
constexpr int kGridBuildThreads = 1024;

struct GridHeader {  // 64 bytes, written by build_grid_kernel
  uint32_t valid, nx, ny, nz;
  float ox, oy, oz, cs;
  float inv_cs, slack, cx, cy;
  float cz, far2;  // grid centre and (5E)^2: rays starting farther away are not admitted
  uint32_t n_big, n_items;
};
static_assert(sizeof(GridHeader) == 64, "GridHeader layout");

// accel buffer: header | big[kGridMaxBig] u16 | cell_start[kGridMaxCells + 2] u16 | items[kGridMaxItems] u16
constexpr size_t kGridBigOff = sizeof(GridHeader);
constexpr size_t kGridStartOff = kGridBigOff + kGridMaxBig * sizeof(uint16_t);
constexpr size_t kGridItemsOff = kGridStartOff + (kGridMaxCells + 2) * sizeof(uint16_t);
constexpr size_t kGridAccelBytes = kGridItemsOff + kGridMaxItems * sizeof(uint16_t);

// LDS image per workgroup: geometry of all n spheres, then the three tables (dword aligned)
__host__ __device__ inline size_t grid_lds_bytes(int n) {
  return (size_t)n * sizeof(float4) + (kGridAccelBytes - kGridBigOff);
}

struct GridLds {
  bool valid;  // wave-uniform
  GridHeader h;
  const float4* geom;
  const uint16_t* big;
  const uint16_t* cell_start;
  const uint16_t* items;
};

__device__ __forceinline__ GridLds stage_grid(const pt_sphere* __restrict__ spheres, int n, const uint32_t* __restrict__ accel,
                                              float4* lds) {
  GridLds g;
  g.h = *reinterpret_cast<const GridHeader*>(accel);
  g.valid = g.h.valid != 0u;
  g.geom = lds;
  uint32_t* tab = reinterpret_cast<uint32_t*>(lds + n);
  g.big = reinterpret_cast<const uint16_t*>(tab);
  g.cell_start = g.big + kGridMaxBig;
  g.items = g.cell_start + (kGridMaxCells + 2);
  if (!g.valid) return g;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const pt_sphere sp = spheres[i];
    lds[i] = make_float4(sp.pos[0], sp.pos[1], sp.pos[2], sp.radius * sp.radius);
  }
  const uint32_t* src = accel + kGridBigOff / 4;
  const int ncells = (int)(g.h.nx * g.h.ny * g.h.nz);
  const int words_a = (kGridMaxBig + ncells + 2 + 1) / 2;  // big list + used part of cell_start
  for (int i = threadIdx.x; i < words_a; i += blockDim.x) tab[i] = src[i];
  const int item_w0 = (int)((kGridItemsOff - kGridBigOff) / 4);
  const int words_i = ((int)g.h.n_items + 1) / 2;
  for (int i = threadIdx.x; i < words_i; i += blockDim.x) tab[item_w0 + i] = src[item_w0 + i];
  __syncthreads();
  return g;
}

// ---- build: one workgroup, everything in LDS -------------------------------------------------------
__device__ __forceinline__ int f2ord(float f) {  // order-preserving float -> int
  const int i = __float_as_int(f);
  return i >= 0 ? i : (int)(0x80000000u - (uint32_t)i);
}
__device__ __forceinline__ float ord2f(int i) { return __int_as_float(i >= 0 ? i : (int)(0x80000000u - (uint32_t)i)); }

__global__ void __launch_bounds__(kGridBuildThreads) build_grid_kernel(const pt_sphere* __restrict__ spheres, int n,
                                                                         uint32_t* __restrict__ accel) {
  __shared__ uint32_t cnt[kGridMaxCells + 1];
  __shared__ uint32_t scan_tmp[kGridBuildThreads];
  __shared__ int bb[6];
  __shared__ float fsum;
  __shared__ uint32_t n_small, n_big, total;
  __shared__ float s_cs;
  __shared__ uint32_t s_dims[3];
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
  // reference radius: geometric mean (a handful of huge walls hardly move it)
  if (tid == 0) {
    fsum = 0.0f;
    n_small = n_big = total = 0u;
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
  const float slack = E * 0.000244140625f;          // 2^-12 E
  const float mk = 36.0f * E * E * 9.5367431640625e-07f;  // 2^-20 (6E)^2: m_j = mk / r_j + slack
  const float r_small = E * 0.001953125f;           // below 2^-9 E the inflation would dwarf the sphere
  auto margin = [&](float r) { return mk / r + slack; };
  auto in_grid = [&](float r) { return r >= r_small && r <= r_big; };
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
    const uint32_t nx = (uint32_t)fminf(ceilf(sx / cs), 1024.0f), ny = (uint32_t)fminf(ceilf(sy / cs), 1024.0f),
                   nz = (uint32_t)fminf(ceilf(sz / cs), 1024.0f);
    const uint32_t ncells = (nx < 1 ? 1 : nx) * (ny < 1 ? 1 : ny) * (nz < 1 ? 1 : nz);
    bool ok = ncells <= (uint32_t)kGridMaxCells;
    if (ok) {
      // count the registrations at this cell size
      __syncthreads();
      for (int c = tid; c <= kGridMaxCells; c += kGridBuildThreads) cnt[c] = 0u;
      if (tid == 0) total = 0u;
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
        for (int z = a[2]; z <= b[2]; z++)
          for (int y = a[1]; y <= b[1]; y++)
            for (int x = a[0]; x <= b[0]; x++) {
              atomicAdd(&cnt[(z * (int)ny + y) * (int)nx + x], 1u);
              lt++;
            }
      }
      atomicAdd(&total, lt);
      __syncthreads();
      ok = total <= (uint32_t)kGridMaxItems;
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
    for (int z = a[2]; z <= b[2]; z++)
      for (int y = a[1]; y <= b[1]; y++)
        for (int x = a[0]; x <= b[0]; x++) {
          const uint32_t p = atomicAdd(&cnt[(z * (int)ny + y) * (int)nx + x], 1u);
          items[p] = (uint16_t)i;
        }
  }
  __syncthreads();
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
    h.cx = 0.5f * (lo[0] + hi[0]);
    h.cy = 0.5f * (lo[1] + hi[1]);
    h.cz = 0.5f * (lo[2] + hi[2]);
    h.far2 = 25.0f * E * E;
    h.n_big = n_big;
    h.n_items = total;
    *hdr = h;
  }
}

// ---- traversal ---------------------------------------------------------------------------------------
struct Near2 {
  float T1, T2;  // two smallest estimates of 2a*t
  int i1;
  bool unsure;
};

// one sphere for one lane: the float part and the estimate of intersect_scene_screened_large, predicated
__device__ __forceinline__ void near2_test(Near2& s, const float4 g, int i, bool en, F3 o, F3 d, const RayConst& rc,
                                           float Tlim_hi) {
  const float INF = __builtin_inff();
  const F3 off = mk3(o.x - g.x, o.y - g.y, o.z - g.z);
  const float b = 2.0f * dot(d, off);
  const float c = dot(off, off) - g.w;
  const float bb = b * b;
  const float a4c = rc.a4 * c;
  const float dacc = fmaf(-rc.a4, c, bb);
  // a sphere can sit in several cells: the current leader must not be entered again as its own runner-up
  const bool cand = en & ((int)__float_as_uint(dacc) >= 0) & !((i == s.i1) & (s.T1 < INF));
  const float sq = __builtin_amdgcn_sqrtf(dacc);
  const float q = b + copysignf(sq, b);
  const float e = fmaf(b, b, -bb);
  const float num = a4c + e;
  const float TA = -q;
  const float TB = -num * __builtin_amdgcn_rcpf(q);
  const float lo = fminf(TA, TB), hi = fmaxf(TA, TB);
  const float T = lo > 0.0f ? lo : hi;
  const bool ok = cand & ((int)__float_as_uint(T) >= 0) & (T < Tlim_hi);
  const float m = fabsf(a4c) * 4.7683716e-07f;  // 2^-21 |4ac|
  s.unsure = s.unsure | (cand & !(fminf(fminf(fabsf(num), fabsf(dacc)), fabsf(a4c)) > m));
  const float Te = ok ? T : INF;
  const bool c1 = Te < s.T1, c2 = Te < s.T2;
  s.T2 = c1 ? s.T1 : (c2 ? Te : s.T2);
  s.i1 = c1 ? i : s.i1;
  s.T1 = c1 ? Te : s.T1;
}

__device__ __forceinline__ bool intersect_scene_grid(const SceneLds& sc, const GridLds& G, int n, F3 o, F3 d, const RayConst& rc,
                                                     float& t_hit, int& idx) {
  const float INF = __builtin_inff();
  const float two_a = 2.0f * rc.a;
  const float Tlim = 1000000.0f * two_a;
  const float Tlim_hi = Tlim * 1.0000153f;
  Near2 s{INF, INF, 0, false};
  // spheres outside the grid (walls, very large or very small ones): every lane tests all of them
  for (int k = 0; k < (int)G.h.n_big; k++) {
    const int i = (int)G.big[k];
    near2_test(s, G.geom[i], i, true, o, d, rc, Tlim_hi);
  }
  const bool admitted = true;  // (the caller has sent the other rays to the brute-force loop)
  // clip against the grid box
  const float gmin[3] = {G.h.ox, G.h.oy, G.h.oz};
  const float dims_f[3] = {(float)G.h.nx, (float)G.h.ny, (float)G.h.nz};
  const int dims[3] = {(int)G.h.nx, (int)G.h.ny, (int)G.h.nz};
  const float oo[3] = {o.x, o.y, o.z}, dd[3] = {d.x, d.y, d.z};
  const float tiny = __builtin_amdgcn_sqrtf(rc.a) * 9.094947e-13f;  // 2^-40 |d|
  float inv[3], t_in = 0.0f, t_out = INF;
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
  const float slack_t = G.h.slack * __builtin_amdgcn_rsqf(rc.a);
  bool active = admitted & (t_in <= t_out + slack_t) & (t_in * two_a < Tlim_hi);
  // entry cell and DDA state
  int cell[3], step[3];
  float tmax[3], tdel[3];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const float p = oo[k] + dd[k] * t_in;
    int ci = (int)floorf((p - gmin[k]) * G.h.inv_cs);
    ci = ci < 0 ? 0 : (ci >= dims[k] ? dims[k] - 1 : ci);
    cell[k] = ci;
    step[k] = par[k] ? 0 : (dd[k] > 0.0f ? 1 : -1);
    const float bnd = gmin[k] + (float)(ci + (dd[k] > 0.0f ? 1 : 0)) * G.h.cs;
    tmax[k] = par[k] ? INF : (bnd - oo[k]) * inv[k];
    tdel[k] = par[k] ? INF : G.h.cs * fabsf(inv[k]);
  }
  int c = (cell[2] * dims[1] + cell[1]) * dims[0] + cell[0];
  c = active ? c : 0;
  uint32_t k0 = G.cell_start[c], k1 = G.cell_start[c + 1];
  if (!active) k1 = k0;
  while (active) {
    if (k0 < k1) {
      const int i = (int)G.items[k0];
      k0++;
      near2_test(s, G.geom[i], i, true, o, d, rc, Tlim_hi);
    }
    if (k0 >= k1) {  // cell finished: leave through the nearest wall
      const float t_exit = fminf(fminf(tmax[0], tmax[1]), tmax[2]);
      const float reach = (t_exit - slack_t) * two_a;
      if ((s.T1 * 1.0000077f < reach) | (reach > Tlim_hi)) {
        active = false;  // nothing that could still matter lies beyond
      } else {
        const int ax = (tmax[0] <= tmax[1]) ? ((tmax[0] <= tmax[2]) ? 0 : 2) : ((tmax[1] <= tmax[2]) ? 1 : 2);
#pragma unroll
        for (int k = 0; k < 3; k++) {
          if (ax == k) {
            cell[k] += step[k];
            tmax[k] += tdel[k];
            if ((cell[k] < 0) | (cell[k] >= dims[k]) | (step[k] == 0)) active = false;
          }
        }
        if (active) {
          c = (cell[2] * dims[1] + cell[1]) * dims[0] + cell[0];
          k0 = G.cell_start[c];
          k1 = G.cell_start[c + 1];
        }
      }
    }
  }
  const bool has = s.T1 < INF;
  bool ambiguous = s.unsure | (has & ((s.T2 <= s.T1 * 1.0000038f) | (s.T1 >= Tlim * 0.99998f)));
  float t;
  bool bad = false;
  const bool real = intersect_sphere_nb(o, d, rc, G.geom[s.i1], t, bad);
  const bool good = real & (t > 0.0f) & (t < 1000000.0f);
  ambiguous = ambiguous | (has & (bad | !good));
  t_hit = t;
  idx = s.i1;
  bool hit = has & good;
  if (__builtin_expect(ambiguous, 0)) hit = intersect_scene_loop<0>(sc, n, o, d, rc, t_hit, idx);
  return hit;
}

// variant 11's nearest-hit search: the grid when the build produced one, the brute-force loop otherwise
__device__ __forceinline__ bool intersect_scene_v11(const SceneLds& sc, int n, F3 o, F3 d, float& t_hit, int& idx) {
  if (n <= 0) return false;
  const RayConst rc = make_ray_const(d);
  const GridLds& G = *sc.grid;
  if (G.valid) {
    // admitted rays only: finite, and starting within 5 E of the grid centre (the registration margins assume it)
    const F3 oc = mk3(o.x - G.h.cx, o.y - G.h.cy, o.z - G.h.cz);
    const float INF = __builtin_inff();
    const bool admitted = (dot(oc, oc) <= G.h.far2) & (rc.a > 0.0f) & (rc.a < 1e30f) & (fabsf(d.x) < INF) & (fabsf(d.y) < INF) &
                          (fabsf(d.z) < INF);
    if (admitted) return intersect_scene_grid(sc, G, n, o, d, rc, t_hit, idx);
  }
  return intersect_scene_screened_large(sc, n, o, d, rc, t_hit, idx);
}

}  // namespace pt
