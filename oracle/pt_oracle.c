/*
 * pt_oracle.c -- CPU ORACLE (test infrastructure, NOT product; see pt_oracle.h header).
 * PARITY STATUS: parity unpinned (no golden vectors in the reference tree; reference not
 * buildable in this image).  Citations are file:line under /root/reference.
 *
 * Build: gcc -std=c99 -O2 -ffp-contract=off -mfma (see oracle/Makefile).  -ffp-contract=off
 * is part of the contract: every '+ - * /' below rounds once, to the type it is written in.
 *
 * ---------------------------------------------------------------------------------------
 * NUMERIC CONTRACT (what "the reference's result" means in this repo)
 * ---------------------------------------------------------------------------------------
 * The reference leaves several things to nvcc / CUDA libm / C++ unspecified behaviour.
 * This oracle fixes them; the HIP kernel implements exactly the same sequence and is
 * required to match bit for bit.
 *  C1. No FMA contraction anywhere the reference writes a*b+c.  Float expressions are
 *      evaluated in float, left to right as C parses them; sub-expressions the reference
 *      promotes to double through 2.0 / 4.0 / 1.0 / 0.2126-style literals are evaluated in
 *      double exactly where C++ promotion rules put them (src/pathtrace.cu:68,75,80-81,134).
 *  C2. helper_math.h normalize(v) = v * rsqrtf(dot(v,v)).  CUDA's device rsqrtf is an
 *      approximation that cannot be reproduced; rsqrtf(x) is DEFINED here as
 *      1.0f / sqrtf(x) (two correctly rounded IEEE operations, = helper_math's host path).
 *  C3. pow(r.y, 1.0f/(power+1.0f)) with power == 1.0f (src/pathtrace.cu:127,133) is
 *      powf(u, 0.5f); DEFINED as sqrtf(u) (correctly rounded).
 *  C4. cos(r.x), sin(r.x) (src/pathtrace.cu:135) on a float argument in (0, 2*pi]:
 *      DEFINED by pto_sincos() below (Cody-Waite reduction + odd/even polynomials using
 *      explicit fmaf; < 1.5 ulp of the true value, checked in tests against libm).
 *  C5. make_float2(curand_uniform(s), curand_uniform(s)) (src/pathtrace.cu:131) has
 *      unspecified argument evaluation order; DEFINED left to right: first draw ->
 *      azimuth r.x, second draw -> elevation r.y.
 *  C6. fminf/fmaxf/min follow IEEE minNum/maxNum; comparisons with NaN are false.  A
 *      negative double discriminant under a non-negative float one yields NaN t values
 *      that the caller's "t > 0" filter rejects -- reproduced, not special-cased.
 *  C7. Images: the reference is square-only (main.cu:66-67) and uses `width` as the row
 *      stride and as the row divisor (pathtrace.cu:206,226).  For W != H this oracle
 *      uses id = row*W + col, sx = row/(float)H, sy = col/(float)W (identical when W==H).
 *  C8. RNG.  xorwow: cuRAND XORWOW restated from the CUDA 8.0 headers from memory
 *      (SURVEY.md section 8(a) row a2), seed = id + p->seed.  philox (this repo's
 *      counter-based mode, no reference counterpart): Philox4x32-10,
 *      key = {seed_lo, seed_hi ^ frame}, ctr = {id, sample, block, 0};
 *      block 0 = {jitter_x, jitter_y, bounce0_az, bounce0_el},
 *      block k>=1 = {bounce(2k-1)_az, _el, bounce(2k)_az, _el}.
 *      Both map u32 -> (0,1] like curand_uniform.
 */
#include "pt_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ float3 helpers ---- */
/* Restated from CUDA samples helper_math.h (third-party, not in /root/reference). */
typedef struct { float x, y, z; } v3;

static inline v3 mk3(float x, float y, float z) { v3 r; r.x = x; r.y = y; r.z = z; return r; }
static inline v3 add3(v3 a, v3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 sub3(v3 a, v3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 mul3(v3 a, v3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 scale3(v3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
static inline float dot3(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline v3 cross3(v3 a, v3 b) {
  return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
/* contract C2 */
static inline v3 normalize3(v3 v) {
  float inv = 1.0f / sqrtf(dot3(v, v));
  return scale3(v, inv);
}
/* lerp(a,b,t) = a + t*(b-a) */
static inline v3 lerp3(v3 a, v3 b, float t) { return add3(a, scale3(sub3(b, a), t)); }
static inline float clampf(float f, float a, float b) { return fmaxf(a, fminf(f, b)); }

/* ------------------------------------------------------------------ RNG --------------- */
/* cuRAND XORWOW, curand_init(seed, 0, 0): include/Renderer.h:37-38, src/pathtrace.cu:265.
 * Third-party (CUDA 8.0 curand_kernel.h), restated from memory, unverified off-NVIDIA. */
void pto_xorwow_init(uint64_t seed, uint32_t st[6]) {
  uint32_t s0 = (uint32_t)seed ^ 0xaad26b49u;
  uint32_t s1 = (uint32_t)(seed >> 32) ^ 0xf7dcefddu;
  uint32_t t0 = 1099087573u * s0;
  uint32_t t1 = 2591861531u * s1;
  st[0] = 6615241u + t1 + t0;   /* d  */
  st[1] = 123456789u + t0;      /* v0 */
  st[2] = 362436069u ^ t0;      /* v1 */
  st[3] = 521288629u + t1;      /* v2 */
  st[4] = 88675123u ^ t1;       /* v3 */
  st[5] = 5783321u + t0;        /* v4 */
}

uint32_t pto_xorwow_next(uint32_t st[6]) {
  uint32_t t = st[1] ^ (st[1] >> 2);
  st[1] = st[2];
  st[2] = st[3];
  st[3] = st[4];
  st[4] = st[5];
  st[5] = (st[5] ^ (st[5] << 4)) ^ (t ^ (t << 1));
  st[0] += 362437u;
  return st[5] + st[0];
}

/* curand_uniform: x * 2^-32 + 2^-33 in float -> (0,1] */
float pto_uniform_from_u32(uint32_t x) {
  return (float)x * 2.3283064e-10f + (2.3283064e-10f / 2.0f);
}

static inline void mulhilo(uint32_t a, uint32_t b, uint32_t* hi, uint32_t* lo) {
  uint64_t p = (uint64_t)a * (uint64_t)b;
  *hi = (uint32_t)(p >> 32);
  *lo = (uint32_t)p;
}

/* Philox4x32-10 (Salmon et al., SC'11); contract C8. */
void pto_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
  uint32_t k0 = key[0], k1 = key[1];
  for (int r = 0; r < 10; r++) {
    uint32_t hi0, lo0, hi1, lo1;
    mulhilo(0xD2511F53u, c0, &hi0, &lo0);
    mulhilo(0xCD9E8D57u, c2, &hi1, &lo1);
    uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

typedef struct {
  int mode;
  uint32_t st[6];                 /* xorwow */
  uint32_t key[2], pix, sample;   /* philox */
  int have_block;
  uint32_t blk[4];
} rng_t;

static void rng_philox_block(rng_t* g, uint32_t block) {
  if (g->have_block != (int)block) {
    uint32_t ctr[4] = { g->pix, g->sample, block, 0u };
    pto_philox4x32_10(ctr, g->key, g->blk);
    g->have_block = (int)block;
  }
}

/* Jitter draws, x then y: src/pathtrace.cu:223-224 */
static void rng_jitter(rng_t* g, float* jx, float* jy) {
  if (g->mode == PTO_RNG_XORWOW) {
    *jx = pto_uniform_from_u32(pto_xorwow_next(g->st));
    *jy = pto_uniform_from_u32(pto_xorwow_next(g->st));
  } else {
    rng_philox_block(g, 0);
    *jx = pto_uniform_from_u32(g->blk[0]);
    *jy = pto_uniform_from_u32(g->blk[1]);
  }
}

/* Bounce draws: src/pathtrace.cu:131, order per contract C5 */
static void rng_bounce(rng_t* g, int n, float* u_az, float* u_el) {
  if (g->mode == PTO_RNG_XORWOW) {
    *u_az = pto_uniform_from_u32(pto_xorwow_next(g->st));
    *u_el = pto_uniform_from_u32(pto_xorwow_next(g->st));
  } else {
    uint32_t block = (n == 0) ? 0u : (uint32_t)((n + 1) >> 1);
    int pair = (n == 0) ? 1 : ((n + 1) & 1);
    rng_philox_block(g, block);
    *u_az = pto_uniform_from_u32(g->blk[2 * pair]);
    *u_el = pto_uniform_from_u32(g->blk[2 * pair + 1]);
  }
}

/* ------------------------------------------------------------------ sincos (C4) ------- */
void pto_sincos(float x, float* s, float* c) {
  /* k = nearest integer to x * 2/pi; x in (0, 2*pi] -> k in 0..4 */
  float kf = rintf(x * 6.366197467e-01f);
  int k = (int)kf;
  /* r = x - k*pi/2, pi/2 split in three floats, each step one fmaf */
  float r = fmaf(-kf, 1.570796371e+00f, x);   /* exact: k <= 4, hi = (float)(pi/2) */
  r = fmaf(-kf, -4.371138829e-08f, r);
  r = fmaf(-kf, -1.715124510e-15f, r);
  float r2 = r * r;
  /* sin(r) = r + r^3 * S(r^2), |r| <= pi/4 (+ rounding slack) */
  float ps = fmaf(r2, 2.755731884e-06f, -1.984127011e-04f);
  ps = fmaf(ps, r2, 8.333333768e-03f);
  ps = fmaf(ps, r2, -1.666666716e-01f);
  float sr = fmaf(r * r2, ps, r);
  /* cos(r) = 1 - r^2/2 + r^4 * C(r^2) */
  float pc = fmaf(r2, -2.755731998e-07f, 2.480158764e-05f);
  pc = fmaf(pc, r2, -1.388888923e-03f);
  pc = fmaf(pc, r2, 4.166666791e-02f);
  float cr = fmaf(r2 * r2, pc, fmaf(r2, -0.5f, 1.0f));
  switch (k & 3) {
    case 0: *s = sr;  *c = cr;  break;
    case 1: *s = cr;  *c = -sr; break;
    case 2: *s = -sr; *c = -cr; break;
    default: *s = -cr; *c = sr; break;
  }
}

/* ------------------------------------------------------------------ scene / camera ---- */
/* include/Scene.h:26-34, values verbatim. */
void pto_scene_cornell(pto_sphere out[9]) {
  static const pto_sphere k[9] = {
    { 1e5f, { 1e5f + 1.0f, 40.8f, 81.6f },      { 0.0f, 0.0f, 0.0f }, { 0.75f, 0.25f, 0.25f } },
    { 1e5f, { -1e5f + 99.0f, 40.8f, 81.6f },    { 0.0f, 0.0f, 0.0f }, { .25f, .25f, .75f } },
    { 1e5f, { 50.0f, 40.8f, 1e5f },             { 0.0f, 0.0f, 0.0f }, { .75f, .75f, .75f } },
    { 1e5f, { 50.0f, 40.8f, -1e5f + 600.0f },   { 0.0f, 0.0f, 0.0f }, { 1.00f, 1.00f, 1.00f } },
    { 1e5f, { 50.0f, 1e5f, 81.6f },             { 0.0f, 0.0f, 0.0f }, { .75f, .75f, .75f } },
    { 1e5f, { 50.0f, -1e5f + 81.6f, 81.6f },    { 0.0f, 0.0f, 0.0f }, { .75f, .75f, .75f } },
    { 16.5f, { 27.0f, 16.5f, 47.0f },           { 0.0f, 0.0f, 0.0f }, { 1.0f, 1.0f, 1.0f } },
    { 16.5f, { 73.0f, 16.5f, 78.0f },           { 0.0f, 0.0f, 0.0f }, { 1.0f, 1.0f, 1.0f } },
    { 600.0f, { 50.0f, 681.6f - .78f, 81.6f },  { 4.0f, 3.6f, 3.2f }, { 0.0f, 0.0f, 0.0f } }
  };
  memcpy(out, k, sizeof(k));
}

/* glm 0.9.8 (third-party, not vendored: README.md:17) restated in float; column-major
 * m[col][row].  include/Camera.h:73-76 (lookAt), :130 (perspective), :131 (inverse). */
typedef struct { float m[4][4]; } m4;

static m4 m4_mul(const m4* a, const m4* b) {
  m4 r;
  for (int j = 0; j < 4; j++)
    for (int i = 0; i < 4; i++)
      r.m[j][i] = a->m[0][i] * b->m[j][0] + a->m[1][i] * b->m[j][1] +
                  a->m[2][i] * b->m[j][2] + a->m[3][i] * b->m[j][3];
  return r;
}

static m4 m4_inverse(const m4* a) {
  /* cofactor expansion on 2x2 sub-determinants */
  const float (*m)[4] = a->m;
  float c00 = m[2][2] * m[3][3] - m[3][2] * m[2][3];
  float c02 = m[1][2] * m[3][3] - m[3][2] * m[1][3];
  float c03 = m[1][2] * m[2][3] - m[2][2] * m[1][3];
  float c04 = m[2][1] * m[3][3] - m[3][1] * m[2][3];
  float c06 = m[1][1] * m[3][3] - m[3][1] * m[1][3];
  float c07 = m[1][1] * m[2][3] - m[2][1] * m[1][3];
  float c08 = m[2][1] * m[3][2] - m[3][1] * m[2][2];
  float c10 = m[1][1] * m[3][2] - m[3][1] * m[1][2];
  float c11 = m[1][1] * m[2][2] - m[2][1] * m[1][2];
  float c12 = m[2][0] * m[3][3] - m[3][0] * m[2][3];
  float c14 = m[1][0] * m[3][3] - m[3][0] * m[1][3];
  float c15 = m[1][0] * m[2][3] - m[2][0] * m[1][3];
  float c16 = m[2][0] * m[3][2] - m[3][0] * m[2][2];
  float c18 = m[1][0] * m[3][2] - m[3][0] * m[1][2];
  float c19 = m[1][0] * m[2][2] - m[2][0] * m[1][2];
  float c20 = m[2][0] * m[3][1] - m[3][0] * m[2][1];
  float c22 = m[1][0] * m[3][1] - m[3][0] * m[1][1];
  float c23 = m[1][0] * m[2][1] - m[2][0] * m[1][1];

  float f0[4] = { c00, c00, c02, c03 }, f1[4] = { c04, c04, c06, c07 };
  float f2[4] = { c08, c08, c10, c11 }, f3[4] = { c12, c12, c14, c15 };
  float f4[4] = { c16, c16, c18, c19 }, f5[4] = { c20, c20, c22, c23 };
  float v0[4] = { m[1][0], m[0][0], m[0][0], m[0][0] };
  float v1[4] = { m[1][1], m[0][1], m[0][1], m[0][1] };
  float v2[4] = { m[1][2], m[0][2], m[0][2], m[0][2] };
  float v3_[4] = { m[1][3], m[0][3], m[0][3], m[0][3] };
  static const float sa[4] = { +1, -1, +1, -1 }, sb[4] = { -1, +1, -1, +1 };
  m4 inv;
  for (int i = 0; i < 4; i++) {
    inv.m[0][i] = (v1[i] * f0[i] - v2[i] * f1[i] + v3_[i] * f2[i]) * sa[i];
    inv.m[1][i] = (v0[i] * f0[i] - v2[i] * f3[i] + v3_[i] * f4[i]) * sb[i];
    inv.m[2][i] = (v0[i] * f1[i] - v1[i] * f3[i] + v3_[i] * f5[i]) * sa[i];
    inv.m[3][i] = (v0[i] * f2[i] - v1[i] * f4[i] + v2[i] * f5[i]) * sb[i];
  }
  float det = (m[0][0] * inv.m[0][0] + m[0][1] * inv.m[1][0]) +
              (m[0][2] * inv.m[2][0] + m[0][3] * inv.m[3][0]);
  float ood = 1.0f / det;
  for (int j = 0; j < 4; j++)
    for (int i = 0; i < 4; i++) inv.m[j][i] *= ood;
  return inv;
}

void pto_camera_basis(const float pos[3], float yaw_deg, float pitch_deg, int w, int h,
                      float basis_out[12]) {
  const float up[3] = {0.0f, 1.0f, 0.0f}; /* Camera.h:58 */
  pto_camera_basis_up(pos, yaw_deg, pitch_deg, up, w, h, basis_out);
}

/* WorldUp explicit: the scalar constructor, Camera.h:63-70 */
void pto_camera_basis_up(const float pos[3], float yaw_deg, float pitch_deg, const float up_in[3], int w, int h,
                         float basis_out[12]) {
  const float deg2rad = 0.01745329251994329576923690768489f;
  /* Camera.h:153-164 updateCameraVectors */
  float yaw = yaw_deg * deg2rad, pitch = pitch_deg * deg2rad;
  v3 front = mk3(cosf(yaw) * cosf(pitch), sinf(pitch), sinf(yaw) * cosf(pitch));
  front = normalize3(front);
  v3 world_up = mk3(up_in[0], up_in[1], up_in[2]);
  v3 right = normalize3(cross3(front, world_up));
  v3 up = normalize3(cross3(right, front));
  v3 eye = mk3(pos[0], pos[1], pos[2]);
  /* Camera.h:73-76 lookAt(Position, Position + Front, Up), right-handed */
  v3 center = add3(eye, front);
  v3 f = normalize3(sub3(center, eye));
  v3 s = normalize3(cross3(f, up));
  v3 u = cross3(s, f);
  m4 view;
  memset(&view, 0, sizeof(view));
  view.m[0][0] = s.x; view.m[1][0] = s.y; view.m[2][0] = s.z;
  view.m[0][1] = u.x; view.m[1][1] = u.y; view.m[2][1] = u.z;
  view.m[0][2] = -f.x; view.m[1][2] = -f.y; view.m[2][2] = -f.z;
  view.m[3][0] = -dot3(s, eye); view.m[3][1] = -dot3(u, eye); view.m[3][2] = dot3(f, eye);
  view.m[3][3] = 1.0f;
  /* Camera.h:130 perspective(radians(45), w/(float)h, 0.01, 1000), RH, z in [-1,1] */
  float fovy = 45.0f * deg2rad, aspect = (float)w / (float)h, zn = 0.01f, zf = 1000.0f;
  float thf = tanf(fovy / 2.0f);
  m4 proj;
  memset(&proj, 0, sizeof(proj));
  proj.m[0][0] = 1.0f / (aspect * thf);
  proj.m[1][1] = 1.0f / thf;
  proj.m[2][2] = -(zf + zn) / (zf - zn);
  proj.m[2][3] = -1.0f;
  proj.m[3][2] = -(2.0f * zf * zn) / (zf - zn);
  /* Camera.h:131 */
  m4 pv = m4_mul(&proj, &view);
  m4 inv = m4_inverse(&pv);
  /* Camera.h:132-148: corners (-1,-1) (+1,-1) (-1,+1) (+1,+1) at NDC z=0, w=1 */
  static const float cx[4] = { -1, +1, -1, +1 }, cy[4] = { -1, -1, +1, +1 };
  for (int k = 0; k < 4; k++) {
    float v[4] = { cx[k], cy[k], 0.0f, 1.0f }, r[4];
    for (int i = 0; i < 4; i++)
      r[i] = (inv.m[0][i] * v[0] + inv.m[1][i] * v[1]) + (inv.m[2][i] * v[2] + inv.m[3][i] * v[3]);
    basis_out[3 * k + 0] = r[0] / r[3] - eye.x;
    basis_out[3 * k + 1] = r[1] / r[3] - eye.y;
    basis_out[3 * k + 2] = r[2] / r[3] - eye.z;
  }
}

/* ------------------------------------------------------------------ the hot path ------ */
/* OnlineVarianceBuffer: src/pathtrace.cu:39-65 */
enum { F_COLOR = 0, F_NORMAL = 1, F_ALBEDO = 2, F_DEPTH = 3, F_NUM = 4 };
typedef struct { int n[F_NUM]; float mean[F_NUM]; float M2[F_NUM]; } varbuf;

static inline void var_update(varbuf* v, float x, int f) {   /* :52-58 */
  v->n[f] += 1;
  float delta = x - v->mean[f];
  v->mean[f] += delta / (float)v->n[f];
  float delta2 = x - v->mean[f];
  v->M2[f] += delta * delta2;
}
static inline float var_get(const varbuf* v, int f) {         /* :60-64 */
  if (v->n[f] < 2) return 0.0f;
  return v->M2[f] / (float)(v->n[f] - 1);
}

/* luminance: src/pathtrace.cu:67-69 -- double literals promote the whole sum (C1) */
static inline float luminance(v3 c) {
  return (float)(0.2126 * (double)c.x + 0.7152 * (double)c.y + 0.0722 * (double)c.z);
}

/* intersectSphere: src/pathtrace.cu:72-91 */
static inline int intersect_sphere(v3 o, v3 d, const pto_sphere* sp, float* t) {
  v3 offset = sub3(o, mk3(sp->pos[0], sp->pos[1], sp->pos[2]));              /* :73 */
  float a = dot3(d, d);                                                      /* :74 */
  float b = (float)(2.0 * (double)dot3(d, offset));                          /* :75 */
  float c = dot3(offset, offset) - sp->radius * sp->radius;                  /* :76 */
  float determinant = b * b - 4.0f * a * c;                                  /* :77 */
  if (determinant >= 0.0f) {                                                 /* :79 */
    double disc = (double)(b * b) - 4.0 * (double)a * (double)c;             /* :80-81 */
    double sq = sqrt(disc);
    float tNear = (float)(((double)(-b) - sq) / (2.0 * (double)a));          /* :80 */
    float tFar = (float)(((double)(-b) + sq) / (2.0 * (double)a));           /* :81 */
    if (tNear > 0.0f && tFar > 0.0f) *t = fminf(tNear, tFar);                /* :82-83 */
    else if (tNear > 0.0f) *t = tNear;                                       /* :84-85 */
    else *t = tFar;                                                          /* :86-87 */
    return 1;
  }
  return 0;
}

int pto_intersect_sphere(const float o[3], const float d[3], const pto_sphere* s, float* t) {
  return intersect_sphere(mk3(o[0], o[1], o[2]), mk3(d[0], d[1], d[2]), s, t);
}

/* intersectScene: src/pathtrace.cu:93-107 */
static inline int intersect_scene(const pto_sphere* sph, int n, v3 o, v3 d, float* t_hit, int* idx) {
  float tNearest = 1000000.0f;
  float t = 0.0f;
  int hit = 0;
  for (int i = 0; i < n; i++) {
    if (intersect_sphere(o, d, &sph[i], &t) && t > 0.0f && t < tNearest) {
      tNearest = t; hit = 1; *t_hit = t; *idx = i;
    }
  }
  return hit;
}

/* orthoVector: src/pathtrace.cu:121-124 */
static inline v3 ortho_vector(v3 v) {
  return (fabsf(v.x) > fabsf(v.z)) ? mk3(-v.y, v.x, 0.0f) : mk3(0.0f, -v.z, v.y);
}

/* getCosineWeightedNormal: src/pathtrace.cu:126-136 (contracts C2-C5) */
static inline v3 cosine_weighted(v3 dir, float u_az, float u_el) {
  dir = normalize3(dir);                                                     /* :128 */
  v3 o1 = normalize3(ortho_vector(dir));                                     /* :129 */
  v3 o2 = normalize3(cross3(dir, o1));                                       /* :130 */
  float rx = u_az * 2.0f * 3.141592654f;                                     /* :132 */
  float ry = sqrtf(u_el);                                                    /* :133, C3 */
  float oneminus = (float)sqrt(1.0 - (double)(ry * ry));                     /* :134 */
  float sn, cs;
  pto_sincos(rx, &sn, &cs);
  v3 a = scale3(o1, cs * oneminus);
  v3 b = scale3(o2, sn * oneminus);
  v3 c = scale3(dir, ry);
  return add3(add3(a, b), c);                                                /* :135 */
}

typedef struct { v3 color, normal, albedo; float depth; } trace_out;

/* trace_ray: src/pathtrace.cu:150-201 */
static void trace_ray(trace_out* L, const pto_sphere* sph, int nsph, v3 o, v3 d, rng_t* g,
                      varbuf* var, int max_bounces) {
  v3 color = mk3(0, 0, 0), mask = mk3(1, 1, 1);
  for (int n = 0; n < max_bounces; n++) {
    float t = 0.0f; int idx = 0;
    if (!intersect_scene(sph, nsph, o, d, &t, &idx)) {                       /* :157-161 */
      L->color = add3(L->color, color);
      return;
    }
    const pto_sphere* s = &sph[idx];
    v3 pos = add3(o, scale3(d, t));                                          /* :163 */
    v3 normal = normalize3(sub3(pos, mk3(s->pos[0], s->pos[1], s->pos[2]))); /* :164 */
    if (!(dot3(normal, d) < 0.0f)) normal = scale3(normal, -1.0f);           /* :166 */
    v3 emis = mk3(s->emission[0], s->emission[1], s->emission[2]);
    v3 scol = mk3(s->color[0], s->color[1], s->color[2]);
    v3 me = mul3(mask, emis);
    if (n == 0)                                                              /* :171-172 */
      color = add3(color, mk3(clampf(me.x, 0.0f, 1.0f), clampf(me.y, 0.0f, 1.0f), clampf(me.z, 0.0f, 1.0f)));
    else                                                                     /* :174 */
      color = add3(color, me);
    mask = mul3(mask, scol);                                                 /* :175 */
    o = add3(pos, scale3(normal, 0.05f));                                    /* :178 */
    float u_az, u_el;
    rng_bounce(g, n, &u_az, &u_el);
    d = normalize3(cosine_weighted(normal, u_az, u_el));                     /* :180 */
    if (n == 0) {                                                            /* :187-195 */
      L->normal = add3(L->normal, normal);
      L->albedo = add3(L->albedo, scol);
      L->depth += t;
      var_update(var, luminance(normal), F_NORMAL);
      var_update(var, luminance(scol), F_ALBEDO);
      var_update(var, t, F_DEPTH);
    }
  }
  L->color = add3(L->color, color);                                          /* :198 */
  var_update(var, luminance(color), F_COLOR);                                /* :200 */
}

/* pixel_kernel body for one pixel: src/pathtrace.cu:203-257 */
static void render_pixel(const pto_params* p, const pto_sphere* sph, int nsph, const v3 B[4], v3 eye,
                         int row, int col, float* out14, uint32_t* state6) {
  rng_t g;
  memset(&g, 0, sizeof(g));
  g.mode = p->rng_mode;
  g.have_block = -1;
  uint32_t id = (uint32_t)row * (uint32_t)p->width + (uint32_t)col;          /* :206 */
  if (g.mode == PTO_RNG_XORWOW) {
    if (state6) memcpy(g.st, state6, sizeof(g.st));                          /* :212 */
    else pto_xorwow_init((uint64_t)id + p->seed, g.st);                      /* :265 */
  } else {
    g.key[0] = (uint32_t)p->seed;
    g.key[1] = (uint32_t)(p->seed >> 32) ^ p->frame;
    g.pix = id;
  }
  varbuf var;
  memset(&var, 0, sizeof(var));
  trace_out L;
  memset(&L, 0, sizeof(L));
  for (int i = 0; i < p->spp; i++) {                                         /* :219 */
    g.sample = (uint32_t)i;
    g.have_block = -1;
    float sx = (float)row, sy = (float)col;                                  /* :221 */
    if (p->spp != 1) {                                                       /* :222-225 */
      float jx, jy;
      rng_jitter(&g, &jx, &jy);
      sx += jx * 1.0f - 0.5f;
      sy += jy * 1.0f - 0.5f;
    }
    sx /= (float)p->height;                                                  /* :226, C7 */
    sy /= (float)p->width;
    v3 dir = lerp3(lerp3(B[0], B[1], sy), lerp3(B[2], B[3], sy), 1.0f - sx); /* :229 */
    trace_ray(&L, sph, nsph, eye, dir, &g, &var, p->max_bounces);            /* :231 */
  }
  float fs = (float)p->spp;                                                  /* :234-237 */
  out14[0] = L.color.x / fs;  out14[1] = L.color.y / fs;  out14[2] = L.color.z / fs;
  out14[3] = L.normal.x / fs; out14[4] = L.normal.y / fs; out14[5] = L.normal.z / fs;
  out14[6] = L.albedo.x / fs; out14[7] = L.albedo.y / fs; out14[8] = L.albedo.z / fs;
  out14[9] = L.depth / fs;
  out14[10] = var_get(&var, F_COLOR);                                        /* :251-254 */
  out14[11] = var_get(&var, F_NORMAL);
  out14[12] = var_get(&var, F_ALBEDO);
  out14[13] = var_get(&var, F_DEPTH);
  if (state6 && g.mode == PTO_RNG_XORWOW) memcpy(state6, g.st, sizeof(g.st)); /* :256 */
}

/* ------------------------------------------------------------------ driver ------------- */
#define PTO_CHUNK 64 /* pixels per work item: fine enough to keep hundreds of host threads busy */
typedef struct {
  const pto_params* p; const pto_sphere* sph; int nsph; v3 B[4]; v3 eye;
  float* out; uint32_t* state; volatile long next_chunk; long n_chunks; long tile_pixels;
} job_t;

static void* worker(void* arg) {
  job_t* j = (job_t*)arg;
  const pto_params* p = j->p;
  for (;;) {
    long ch = __sync_fetch_and_add(&j->next_chunk, 1L);
    if (ch >= j->n_chunks) break;
    long tp_end = (ch + 1) * PTO_CHUNK < j->tile_pixels ? (ch + 1) * PTO_CHUNK : j->tile_pixels;
    for (long tp = ch * PTO_CHUNK; tp < tp_end; tp++) {
      int row = p->row_begin + (int)(tp / p->width), col = (int)(tp % p->width);
      render_pixel(p, j->sph, j->nsph, j->B, j->eye, row, col, j->out + (size_t)tp * 14,
                   j->state ? j->state + (size_t)tp * 6 : NULL);
    }
  }
  return NULL;
}

static int params_ok(const pto_params* p) {
  return p && p->width > 0 && p->height > 0 && p->row_begin >= 0 && p->row_end <= p->height &&
         p->row_begin <= p->row_end && p->spp > 0 && p->max_bounces >= 0 &&
         (p->rng_mode == PTO_RNG_XORWOW || p->rng_mode == PTO_RNG_PHILOX);
}

int pto_render(const pto_params* p, const pto_sphere* spheres, int n_spheres, const float basis[12],
               const float eye[3], float* out, uint32_t* rng_state, int n_threads) {
  if (!params_ok(p) || n_spheres < 0 || (n_spheres > 0 && !spheres) || !basis || !eye || !out) return -1;
  job_t j;
  j.p = p; j.sph = spheres; j.nsph = n_spheres; j.out = out; j.state = rng_state;
  for (int k = 0; k < 4; k++) j.B[k] = mk3(basis[3 * k], basis[3 * k + 1], basis[3 * k + 2]);
  j.eye = mk3(eye[0], eye[1], eye[2]);
  j.tile_pixels = (long)(p->row_end - p->row_begin) * (long)p->width;
  j.n_chunks = (j.tile_pixels + PTO_CHUNK - 1) / PTO_CHUNK;
  j.next_chunk = 0;
  if (n_threads < 1) n_threads = 1;
  if (n_threads > 256) n_threads = 256;
  if (n_threads == 1) { worker(&j); return 0; }
  pthread_t th[256];
  int started = 0;
  for (int i = 0; i < n_threads; i++)
    if (pthread_create(&th[started], NULL, worker, &j) == 0) started++;
  if (started == 0) worker(&j);
  for (int i = 0; i < started; i++) pthread_join(th[i], NULL);
  return 0;
}

/* denoise_kernel: src/denoise.cu:9-29.  min/max are CUDA's float overloads (fminf/fmaxf: a NaN
 * operand yields the other one), colour*255.0 is a double product truncated to unsigned char. */
void pto_display_pack(const float* in, int width, int height, float* out) {
  for (int x = 0; x < height; x++)        /* x = row, :10 */
    for (int y = 0; y < width; y++) {     /* y = col, :11 */
      const float* px = in + ((size_t)x * width + y) * 14;           /* :16 */
      unsigned char c[4];
      for (int k = 0; k < 3; k++) {
        float v = fminf(fmaxf(px[k], 0.0f), 1.0f);                   /* :18-20 */
        c[k] = (unsigned char)((double)v * 255.0);                   /* :23 */
      }
      c[3] = 1;
      float packed;
      memcpy(&packed, c, 4);                                         /* union Color, :3-7 */
      float* o = out + ((size_t)x * width + y) * 3;
      o[0] = (float)y;                                               /* :26 */
      o[1] = (float)(width - x);                                     /* :27 */
      o[2] = packed;                                                 /* :28 */
    }
}

void pto_setup_random(const pto_params* p, uint32_t* rng_state) {
  for (int row = p->row_begin; row < p->row_end; row++)
    for (int col = 0; col < p->width; col++) {
      uint32_t id = (uint32_t)row * (uint32_t)p->width + (uint32_t)col;
      size_t tp = (size_t)(row - p->row_begin) * (size_t)p->width + (size_t)col;
      pto_xorwow_init((uint64_t)id + p->seed, rng_state + tp * 6);
    }
}
