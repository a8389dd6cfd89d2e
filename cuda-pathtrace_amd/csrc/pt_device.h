// pt_device.h -- device-side building blocks of the gfx950 path-trace megakernel.
//
// Numeric contract (DESIGN.md "Numeric contract"): every float / double operation below is
// a single correctly rounded IEEE operation, written in the order the reference writes it
// (src/pathtrace.cu, cited per function).  This translation unit is compiled with
// -ffp-contract=off; the only fused operations are the explicit fmaf() calls of pt_sincos,
// which is this repo's definition of the reference's device sinf/cosf.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#pragma clang fp contract(off)

namespace pt {

struct F3 {
  float x, y, z;
};

__device__ __forceinline__ F3 mk3(float x, float y, float z) { return F3{x, y, z}; }

// v_bitop3_b32: ANY bitwise function of three words in one instruction, and the only three-operand bit instruction that issues at
// the FULL rate on gfx950 (2.2 cycles per wave64; v_bfi_b32, v_and_or_b32, v_or3_b32, every shift, v_cndmask_b32: 4.1 --
// tools/ubench/valu_clock.hip, profiles/r03/valu_clock_bit_ops.txt).  TT is the truth table with a = 0xF0, b = 0xCC, c = 0xAA,
// e.g. (a & c) | (b & ~c) = 0xE4.  The compiler picks it for some and/or/xor trees and v_bfi/v_and_or for others, so the
// hot path asks for it by name.
template <int TT>
__device__ __forceinline__ uint32_t bitop3(uint32_t a, uint32_t b, uint32_t c) {
  return __builtin_amdgcn_bitop3_b32(a, b, c, TT);
}
// A constant that must live in a VGPR.  ANY VALU instruction with an SGPR source operand issues at half rate on gfx950 -- even
// v_add_f32 v, s, v (tools/ubench/valu_banks.hip, profiles/r03/valu_banks.txt) -- and the compiler keeps the non-inline constants
// of VOP3 instructions in SGPRs.  The move is opaque to it.  Called ONCE per kernel (stage_scene: SceneLds::absmask) and handed
// down: the unroller prices every inline-assembly statement highly, and one per sphere test stops the bounce loop from being
// unrolled.  Inline constants (-16..64, 0.5, 1, 2, 4 and their negatives) cost nothing either way.
template <uint32_t C>
__device__ __forceinline__ uint32_t vgpr_const() {
  uint32_t r;
  asm("v_mov_b32 %0, %1" : "=v"(r) : "n"(C));
  return r;
}
// copysignf(mag, -sgn): (a & c) | (~b & ~c); absmask = 0x7FFFFFFF, in a VGPR for the hot paths (SceneLds::absmask)
__device__ __forceinline__ float copysign_neg_b3(float mag, float sgn, uint32_t absmask = 0x7FFFFFFFu) {
  return __uint_as_float(bitop3<0xB1>(__float_as_uint(mag), __float_as_uint(sgn), absmask));
}
// copysignf(mag, sgn) (the library form compiles to the half-rate v_bfi_b32)
__device__ __forceinline__ float copysign_b3(float mag, float sgn) {
  return __uint_as_float(bitop3<0xE4>(__float_as_uint(mag), __float_as_uint(sgn), 0x7FFFFFFFu));
}
__device__ __forceinline__ F3 operator+(F3 a, F3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ F3 operator-(F3 a, F3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ F3 operator*(F3 a, F3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ F3 operator*(F3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ float dot(F3 a, F3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ F3 cross(F3 a, F3 b) {
  return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
// helper_math.h normalize(v) = v * rsqrtf(dot(v,v)); rsqrtf(x) := 1.0f / sqrtf(x) (contract C2)
__device__ __forceinline__ F3 normalize(F3 v) {
  float inv = 1.0f / sqrtf(dot(v, v));
  return v * inv;
}
__device__ __forceinline__ F3 lerp(F3 a, F3 b, float t) { return a + (b - a) * t; }
__device__ __forceinline__ float clampf(float f, float a, float b) { return fmaxf(a, fminf(f, b)); }

// ---- RNG ---------------------------------------------------------------------------------
// cuRAND XORWOW (curand_init(seed,0,0) / curand / curand_uniform): src/pathtrace.cu:131,
// 223-224,265.  Six 32-bit words live in VGPRs for the whole kernel.
struct Xorwow {
  uint32_t d, v0, v1, v2, v3, v4;
};

__device__ __forceinline__ void xorwow_init(Xorwow& s, uint64_t seed) {
  uint32_t s0 = (uint32_t)seed ^ 0xaad26b49u;
  uint32_t s1 = (uint32_t)(seed >> 32) ^ 0xf7dcefddu;
  uint32_t t0 = 1099087573u * s0;
  uint32_t t1 = 2591861531u * s1;
  s.d = 6615241u + t1 + t0;
  s.v0 = 123456789u + t0;
  s.v1 = 362436069u ^ t0;
  s.v2 = 521288629u + t1;
  s.v3 = 88675123u ^ t1;
  s.v4 = 5783321u + t0;
}

__device__ __forceinline__ uint32_t xorwow_next(Xorwow& s) {
  uint32_t t = s.v0 ^ (s.v0 >> 2);
  s.v0 = s.v1;
  s.v1 = s.v2;
  s.v2 = s.v3;
  s.v3 = s.v4;
  // (v4 ^ (v4 << 4)) ^ (t ^ (t << 1)): the same word from one left shift (half rate; right shifts are full rate), one add
  // (t << 1 = t + t; written as the instruction, the compiler turns the sum back into a shift) and one three-way xor
  uint32_t t2;
  asm("v_add_u32 %0, %1, %1" : "=v"(t2) : "v"(t));
  s.v4 = bitop3<0x96>(s.v4 ^ (s.v4 << 4), t, t2);
  s.d += 362437u;
  return s.v4 + s.d;
}

// Advance the generator by n draws without producing outputs (variant 8's skip-ahead).  The
// xorshift part rotates five words, so five steps written out need no register moves; the Weyl
// counter d advances in one multiply.
__device__ __forceinline__ void xorwow_skip(Xorwow& s, int n) {
  s.d += 362437u * (uint32_t)n;
  uint32_t v0 = s.v0, v1 = s.v1, v2 = s.v2, v3 = s.v3, v4 = s.v4;
#define PT_XW_STEP(a, e) { const uint32_t t = (a) ^ ((a) >> 2); (a) = ((e) ^ ((e) << 4)) ^ (t ^ (t << 1)); }
  // after a step the new v4 lives in the register that held v0: (v0,v1,v2,v3,v4) -> (v1,v2,v3,v4,new)
  for (; n >= 5; n -= 5) {
    PT_XW_STEP(v0, v4)  // new word in v0, logical order now v1 v2 v3 v4 v0
    PT_XW_STEP(v1, v0)
    PT_XW_STEP(v2, v1)
    PT_XW_STEP(v3, v2)
    PT_XW_STEP(v4, v3)  // logical order back to v0 v1 v2 v3 v4
  }
  for (; n > 0; n--) {
    PT_XW_STEP(v0, v4)
    const uint32_t nw = v0;
    v0 = v1; v1 = v2; v2 = v3; v3 = v4; v4 = nw;
  }
#undef PT_XW_STEP
  s.v0 = v0; s.v1 = v1; s.v2 = v2; s.v3 = v3; s.v4 = v4;
}

// the same for a step count known at compile time, straight-line: no loop, no trip-count arithmetic; the rotation the last
// N % 5 steps leave is resolved by the register allocator instead of by moves
template <int N>
__device__ __forceinline__ void xorwow_skip_n(Xorwow& s) {
  s.d += 362437u * (uint32_t)N;
  uint32_t v[5] = {s.v0, s.v1, s.v2, s.v3, s.v4};
#pragma unroll
  for (int k = 0; k < N; k++) {  // step k reads the oldest word v[k % 5] and the newest v[(k + 4) % 5], and replaces the oldest
    const uint32_t a = v[k % 5], e = v[(k + 4) % 5];
    const uint32_t t = a ^ (a >> 2);
    v[k % 5] = (e ^ (e << 4)) ^ (t ^ (t << 1));
  }
  s.v0 = v[N % 5]; s.v1 = v[(N + 1) % 5]; s.v2 = v[(N + 2) % 5]; s.v3 = v[(N + 3) % 5]; s.v4 = v[(N + 4) % 5];
}

// curand_uniform: (0,1] -- _curand_uniform's x * 2^-32 + 2^-33 (contract C5).  The product with a power of two is exact, so
// the one fma below rounds the same exact sum the multiply-then-add form rounds: identical for every x (and checked for all
// 2^32 of them, tests/test_unary_exhaustive_gpu.py), one instruction less per draw.
__device__ __forceinline__ float uniform_from_u32_literal(uint32_t x) {
  return (float)x * 2.3283064e-10f + (2.3283064e-10f / 2.0f);
}
__device__ __forceinline__ float uniform_from_u32(uint32_t x) {
  return fmaf((float)x, 2.3283064e-10f, 2.3283064e-10f / 2.0f);
}

// Philox4x32-10, counter-based (contract C8): nothing is kept between samples.
__device__ __forceinline__ uint4 philox4x32_10(uint4 c, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; r++) {
    uint32_t hi0 = __umulhi(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
    uint32_t hi1 = __umulhi(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
    c = make_uint4(hi1 ^ c.y ^ k0, lo1, hi0 ^ c.w ^ k1, lo0);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return c;
}

// ---- sin/cos on (0, 2*pi] (contract C4) ----------------------------------------------------
// The definition, as the oracle writes it (oracle/pt_oracle.c, pto_sincos): rintf, int conversion, compares and selects.
__device__ __forceinline__ void pt_sincos_literal(float x, float& s, float& c) {
  float kf = rintf(x * 6.366197467e-01f);
  int k = (int)kf;
  float r = fmaf(-kf, 1.570796371e+00f, x);
  r = fmaf(-kf, -4.371138829e-08f, r);
  r = fmaf(-kf, -1.715124510e-15f, r);
  float r2 = r * r;
  float ps = fmaf(r2, 2.755731884e-06f, -1.984127011e-04f);
  ps = fmaf(ps, r2, 8.333333768e-03f);
  ps = fmaf(ps, r2, -1.666666716e-01f);
  float sr = fmaf(r * r2, ps, r);
  float pc = fmaf(r2, -2.755731998e-07f, 2.480158764e-05f);
  pc = fmaf(pc, r2, -1.388888923e-03f);
  pc = fmaf(pc, r2, 4.166666791e-02f);
  float cr = fmaf(r2 * r2, pc, fmaf(r2, -0.5f, 1.0f));
  bool swap = (k & 1) != 0;
  float ss = swap ? cr : sr;
  float cc = swap ? sr : cr;
  // quadrant signs: k&3 = 0:(s,c) 1:(c,-s) 2:(-s,-c) 3:(-c,s)
  s = (k & 2) ? -ss : ss;
  c = ((k + 1) & 2) ? -cc : cc;
}
// The same two floats for every |x| < 2^22 * pi/2 except x = -0 (tests/test_unary_exhaustive_gpu.py compares all of those bit
// patterns; the path's argument is u * 2 * pi in (0, 2*pi]) from full-rate instructions only, bar one shift: the rounding to
// the nearest quadrant is the magic-number addition (t + 1.5 * 2^23 rounds t to an integer, ties to even, exactly as rintf does,
// and leaves k in the sum's low mantissa bits: no v_rndne_f32, no v_cvt_i32_f32), the quadrant's exchange and signs are bit
// selections and sign flips (v_bitop3_b32) instead of three compares and four selects at half rate: 48 -> 28 issue cycles.
__device__ __forceinline__ void pt_sincos(float x, float& s, float& c, uint32_t absmask = 0x7FFFFFFFu) {
  const float t = x * 6.366197467e-01f;
  const float y = t + 12582912.0f;
  const float kf = y - 12582912.0f;
  const uint32_t kb = __float_as_uint(y);  // 0x4B400000 + k: the low bits are k's (two's complement)
  float r = fmaf(-kf, 1.570796371e+00f, x);
  r = fmaf(-kf, -4.371138829e-08f, r);
  r = fmaf(-kf, -1.715124510e-15f, r);
  float r2 = r * r;
  float ps = fmaf(r2, 2.755731884e-06f, -1.984127011e-04f);
  ps = fmaf(ps, r2, 8.333333768e-03f);
  ps = fmaf(ps, r2, -1.666666716e-01f);
  float sr = fmaf(r * r2, ps, r);
  float pc = fmaf(r2, -2.755731998e-07f, 2.480158764e-05f);
  pc = fmaf(pc, r2, -1.388888923e-03f);
  pc = fmaf(pc, r2, 4.166666791e-02f);
  float cr = fmaf(r2 * r2, pc, fmaf(r2, -0.5f, 1.0f));
  const uint32_t odd = 0u - (kb & 1u);  // all ones in an odd quadrant: sine and cosine change places
  const uint32_t ss = bitop3<0xE4>(__float_as_uint(cr), __float_as_uint(sr), odd);  // (a & c) | (b & ~c)
  const uint32_t cc = bitop3<0xE4>(__float_as_uint(sr), __float_as_uint(cr), odd);
  const uint32_t w = kb << 30;       // bit 31 = bit 1 of k: the sine's sign
  const uint32_t w1 = w ^ (w + w);   // bit 31 = bit 1 ^ bit 0 of k = bit 1 of k + 1: the cosine's sign
  s = __uint_as_float(bitop3<0xB4>(ss, w, absmask));  // a ^ (b & ~c): the sign bit of w flips ss (absmask in a VGPR: hot paths)
  c = __uint_as_float(bitop3<0xB4>(cc, w1, absmask));
}

// ---- luminance: src/pathtrace.cu:67-69 (double through the literals) -----------------------
__device__ __forceinline__ float luminance(F3 c) {
  return (float)(0.2126 * (double)c.x + 0.7152 * (double)c.y + 0.0722 * (double)c.z);
}

// ---- OnlineVarianceBuffer: src/pathtrace.cu:39-65 -------------------------------------------
struct Welford {
  int n;
  float mean, M2;
};
__device__ __forceinline__ void welford_update(Welford& w, float x) {
  w.n += 1;
  float delta = x - w.mean;
  w.mean += delta / (float)w.n;
  float delta2 = x - w.mean;
  w.M2 += delta * delta2;
}

// delta / (float)n of the update above (:52) without the division, n = 1, 2, ...  With y = the correctly rounded 1/(float)n
// (SceneLds::rcpn: a per-workgroup LDS table filled by the division itself), q0 = delta * y, the exact remainder
// r = delta - n * q0 and one correction give the correctly rounded quotient (Markstein) whenever q0 is a normal number or
// delta is +0: a correctly rounded division is ~13 instructions, this is a multiply, two fmas and two integer range tests.  Not argued but
// CHECKED: tests/test_unary_exhaustive_gpu.py compares it with the division for every one of the 2^32 float deltas and every
// n of the table.  Anything else (q0 subnormal, zero, inf, NaN -- the table's last entry is a NaN, which is what n beyond
// the table reads) takes the division.
constexpr int kRcpTab = 1024;
__device__ __forceinline__ float div_by_count(float delta, float nf, float y) {
  const float q0 = delta * y;
  const float r = fmaf(-nf, q0, delta);
  float q = fmaf(r, y, q0);
  // valid when q0 is a normal number (the remainder is then exact) or the dividend is +0 -- which it is whenever a sample
  // equals the running mean, every sample of a pixel that sees one flat-shaded surface.  Integer tests on the bit patterns:
  // __builtin_amdgcn_class(q0, normal) was seen to answer true for denormal products here (tools/ubench/div_probe.hip).
  // (q0 enters the zero test so that the NaN reciprocal of a count beyond the table sends a zero dividend to the division too)
  const bool ok = (((__float_as_uint(q0) & 0x7F800000u) - 0x00800000u) < 0x7F000000u) | ((__float_as_uint(delta) | __float_as_uint(q0)) == 0u);
  if (__builtin_expect(!ok, 0)) q = delta / nf;
  return q;
}
__device__ __forceinline__ float rcp_count(const float* __restrict__ rcpn, int n) {
  const uint32_t k = (uint32_t)n - 1u;
  return rcpn[k < (uint32_t)kRcpTab ? k : (uint32_t)kRcpTab];
}
__device__ __forceinline__ void welford_update(Welford& w, float x, const float* __restrict__ rcpn) {
  w.n += 1;
  const float delta = x - w.mean;
  w.mean += div_by_count(delta, (float)w.n, rcp_count(rcpn, w.n));
  const float delta2 = x - w.mean;
  w.M2 += delta * delta2;
}
// the three first-hit accumulators (:187-195) are updated together or not at all: one count, one reciprocal
__device__ __forceinline__ void welford_update3(Welford& a, Welford& b, Welford& c, float xa, float xb, float xc,
                                                const float* __restrict__ rcpn) {
  a.n += 1;
  b.n = a.n;
  c.n = a.n;
  const float nf = (float)a.n, y = rcp_count(rcpn, a.n);
  const float da = xa - a.mean, db = xb - b.mean, dc = xc - c.mean;
  a.mean += div_by_count(da, nf, y);
  b.mean += div_by_count(db, nf, y);
  c.mean += div_by_count(dc, nf, y);
  a.M2 += da * (xa - a.mean);
  b.M2 += db * (xb - b.mean);
  c.M2 += dc * (xc - c.mean);
}
__device__ __forceinline__ float welford_variance(const Welford& w) {
  return (w.n < 2) ? 0.0f : w.M2 / (float)(w.n - 1);
}

// ---- orthoVector / getCosineWeightedNormal: src/pathtrace.cu:121-136 -------------------------
__device__ __forceinline__ F3 ortho_vector(F3 v) {
  return (fabsf(v.x) > fabsf(v.z)) ? mk3(-v.y, v.x, 0.0f) : mk3(0.0f, -v.z, v.y);
}

__device__ __forceinline__ F3 cosine_weighted(F3 dir, float u_az, float u_el) {
  dir = normalize(dir);
  F3 o1 = normalize(ortho_vector(dir));
  F3 o2 = normalize(cross(dir, o1));
  float rx = u_az * 2.0f * 3.141592654f;
  float ry = sqrtf(u_el);  // powf(u, 0.5f) := sqrtf(u), contract C3
  float oneminus = (float)sqrt(1.0 - (double)(ry * ry));
  float sn, cs;
  pt_sincos_literal(rx, sn, cs);  // the literal kernel keeps the definition
  F3 a = o1 * (cs * oneminus);
  F3 b = o2 * (sn * oneminus);
  F3 c = dir * ry;
  return (a + b) + c;
}

// ---- intersectSphere: src/pathtrace.cu:72-91 -------------------------------------------------
// g = {centre.xyz, radius*radius}; a = dot(d,d) is loop-invariant over the sphere list and
// passed in.  Returns the reference's `*t` through t and its bool through the return value.
__device__ __forceinline__ bool intersect_sphere(F3 o, F3 d, float a, float4 g, float& t) {
  F3 off = mk3(o.x - g.x, o.y - g.y, o.z - g.z);
  float b = 2.0f * dot(d, off);  // == (float)(2.0 * (double)dot): scaling by 2 is exact
  float c = dot(off, off) - g.w;
  float bb = b * b;
  float det = bb - 4.0f * a * c;
  if (det >= 0.0f) {
    double disc = (double)bb - 4.0 * (double)a * (double)c;
    double sq = sqrt(disc);
    double den = 2.0 * (double)a;
    float tn = (float)(((double)(-b) - sq) / den);
    float tf = (float)(((double)(-b) + sq) / den);
    if (tn > 0.0f && tf > 0.0f)
      t = fminf(tn, tf);
    else if (tn > 0.0f)
      t = tn;
    else
      t = tf;
    return true;
  }
  return false;
}

// ---- cheap correctly rounded 1/sqrt and sqrt (float) ------------------------------------------
// normalize() needs inv = RN(1.0f / RN(sqrtf(x))).  hipcc expands the two correctly rounded
// operations into ~24 instructions (range scaling, v_sqrt + two residual fixups with selects,
// v_div_scale/v_rcp/4 fma/v_div_fmas/v_div_fixup).  For x in [2^-100, 2^100] the sequences
// below give the same bits from 7 (5) instructions: one Newton step on v_rsq_f32's seed with an
// fma residual lands on the correctly rounded sqrt, and one more lands on the correctly rounded
// reciprocal of THAT sqrt (plus one ulp when the sqrt's significand is all ones, see below).
// This is not argued, it is CHECKED: tests/test_unary_exhaustive_gpu.py
// compares them with the literal expressions for every one of the 2^32 float bit patterns on
// the GPU (0 mismatches), and the literal device expressions with the CPU on a dense sample.
// Outside the range (zero, denormal, huge, inf, NaN, negative) the literal code runs.
__device__ __forceinline__ bool in_fast_range(float x) {
#ifdef PT_TIMING_ONLY_NO_FALLBACKS  // never defined in a shipped build: upper bound of what the rare branches cost
  return true;
#endif
  return (__float_as_uint(x) - 0x0D800000u) < 0x64000000u;  // sign clear and 2^-100 <= x < 2^100 (exponent field 27..226)
}

__device__ __forceinline__ float sqrt_cr_f32(float x) {
  if (__builtin_expect(!in_fast_range(x), 0)) return sqrtf(x);
  const float y = __builtin_amdgcn_rsqf(x);
  const float s0 = x * y;
  const float h = 0.5f * y;
  const float r = fmaf(-s0, s0, x);
  return fmaf(r, h, s0);
}

__device__ __forceinline__ float inv_sqrt_spec(float x) {
  if (__builtin_expect(!in_fast_range(x), 0)) return 1.0f / sqrtf(x);
  const float y = __builtin_amdgcn_rsqf(x);
  const float s0 = x * y;
  const float h = 0.5f * y;
  const float r = fmaf(-s0, s0, x);
  const float s1 = fmaf(r, h, s0);
  const float e = fmaf(-s1, y, 1.0f);
  const float inv = fmaf(e, y, y);
  // 1/s1 for an all-ones significand (s1 = 2^k (1 - 2^-24), e.g. the sqrt of the squared length
  // 1 - 2^-24 of an already normalised vector -- common!) lies just above a rounding tie: the
  // Newton step returns 2^-k, the correctly rounded value is one ulp up.  Found, and the fix
  // verified for all 2^32 inputs, by the exhaustive test.
  return __uint_as_float(__float_as_uint(inv) + (((__float_as_uint(s1) & 0x7FFFFFu) == 0x7FFFFFu) ? 1u : 0u));
}

__device__ __forceinline__ F3 normalize_fast(F3 v) { return v * inv_sqrt_spec(dot(v, v)); }

// ---- variant 1: the same values from fewer FP64 instructions -----------------------------------
// Everything below returns bit-for-bit what intersect_sphere() returns; only the instruction
// sequence differs.  Per bounce the ray direction is fixed, so den = 2.0*a and an almost
// correctly rounded reciprocal of it are computed once (RayConst) and shared by all spheres.
struct RayConst {
  float a;      // dot(d,d)
  float a4;     // 4.0f*a (exact)
  double a4d;   // 4.0*(double)a (exact)
  double den;   // 2.0*(double)a (exact)
  double rden;  // ~1/den, <= 1 ulp
};

// ~1/den to <= 1 ulp: v_rcp_f64 and two Newton steps
__device__ __forceinline__ double ray_const_rden(double den) {
  double r = __builtin_amdgcn_rcp(den);
  double e = __builtin_fma(-den, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-den, r, 1.0);
  return __builtin_fma(r, e, r);
}

__device__ __forceinline__ RayConst make_ray_const(F3 d) {
  RayConst rc;
  rc.a = dot(d, d);
  rc.a4 = 4.0f * rc.a;
  rc.a4d = 4.0 * (double)rc.a;
  rc.den = 2.0 * (double)rc.a;
  rc.rden = ray_const_rden(rc.den);
  return rc;
}

// make_ray_const for a direction that is of unit length up to rounding (every ray but the primary one: bounce_geometry
// normalises it): a = dot(d, d) is within a few ulps of 1, so the refined reciprocal of 2a is one of a handful of doubles,
// tabulated once per workgroup by make_ray_const itself (SceneLds::rden1, same index as normalize_unit_nb's table).  Saves
// the v_rcp_f64 and its four refinement steps per bounce.  Any other a takes the general routine.
__device__ __forceinline__ RayConst make_ray_const_unit(F3 d, const double* __restrict__ tab) {
  RayConst rc;
  rc.a = dot(d, d);
  const uint32_t k = __float_as_uint(rc.a) - (0x3F800000u - 16u);
  if (__builtin_expect(k > 32u, 0)) return make_ray_const(d);
  rc.a4 = 4.0f * rc.a;
  rc.a4d = 4.0 * (double)rc.a;
  rc.den = 2.0 * (double)rc.a;
  rc.rden = tab[k & 63u];
  return rc;
}

// Correctly rounded sqrt for x in [2^-600, 2^600]: the Newton core hipcc itself emits for
// sqrt(double) (v_rsq_f64 + Goldschmidt/Newton with fma residuals) without its range
// scaling; anything else (zero, negative, tiny, huge, inf, NaN) takes the library call.
__device__ __forceinline__ double sqrt_cr(double x) {
  const uint32_t hi = (uint32_t)__double2hiint(x);
#ifndef PT_TIMING_ONLY_NO_FALLBACKS
  if (__builtin_expect(((hi >> 20) - 423u) >= 1200u, 0)) return sqrt(x);
#endif
  double y = __builtin_amdgcn_rsq(x);
  double g = x * y;
  double h = y * 0.5;
  double r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g);
  h = __builtin_fma(h, r, h);
  double d = __builtin_fma(-g, g, x);
  g = __builtin_fma(d, h, g);
  d = __builtin_fma(-g, g, x);
  g = __builtin_fma(d, h, g);
  return g;
}

// (float)(num / den) with den = rc.den.  q = Markstein quotient from the shared reciprocal is
// within 1 ulp(double) of the correctly rounded quotient; the float it rounds to can differ
// from the reference's only if q lies within a few double-ulps of a float rounding boundary
// (low 29 mantissa bits == 0x10000000) or outside the float normal range -- those lanes
// (about 3e-8 of all quotients) redo the division literally.
__device__ __forceinline__ float quotient_to_float(double num, const RayConst& rc) {
  double q = num * rc.rden;
  double rem = __builtin_fma(-q, rc.den, num);
  q = __builtin_fma(rem, rc.rden, q);
  const uint32_t lo = (uint32_t)__double2loint(q), hi = (uint32_t)__double2hiint(q);
  const bool near_boundary = ((lo & 0x1FFFFFFFu) - 0x0FFFFFF8u) <= 0x10u;
  const bool out_of_range = (((hi >> 20) & 0x7FFu) - 903u) >= 247u;
#ifndef PT_TIMING_ONLY_NO_FALLBACKS
  if (__builtin_expect(near_boundary || out_of_range, 0)) q = num / rc.den;
#endif
  return (float)q;
}

__device__ __forceinline__ bool intersect_sphere_v1(F3 o, F3 d, const RayConst& rc, float4 g, float& t) {
  F3 off = mk3(o.x - g.x, o.y - g.y, o.z - g.z);
  float b = 2.0f * dot(d, off);
  float c = dot(off, off) - g.w;
  float bb = b * b;
  float det = bb - rc.a4 * c;
  if (det >= 0.0f) {
    double disc = (double)bb - rc.a4d * (double)c;
    double sq = sqrt_cr(disc);
    double nb = (double)(-b);
    float tn = quotient_to_float(nb - sq, rc);
    float tf = quotient_to_float(nb + sq, rc);
    if (tn > 0.0f && tf > 0.0f)
      t = fminf(tn, tf);
    else if (tn > 0.0f)
      t = tn;
    else
      t = tf;
    return true;
  }
  return false;
}

// ---- branch-free forms (variant 6) ------------------------------------------------------------
// Same sequences, but instead of branching to the literal code on the rare inputs they cannot
// handle, they OR a flag; the caller redoes the whole step literally when the flag is set.
// One deferred branch per step instead of ~15 tiny ones per bounce keeps the code straight-line
// (bigger scheduling regions, fewer scalar branch instructions), which matters most when a
// small tile leaves only two waves per SIMD.
// CHECKED = false: the caller knows x to be inside the verified range (2^-100 .. 2^100), see bounce_geometry
template <bool CHECKED = true>
__device__ __forceinline__ float sqrt_cr_f32_nb(float x, bool& bad) {
  if constexpr (CHECKED) bad = bad | !in_fast_range(x);
  const float y = __builtin_amdgcn_rsqf(x);
  const float s0 = x * y;
  const float h = 0.5f * y;
  const float r = fmaf(-s0, s0, x);
  return fmaf(r, h, s0);
}

template <bool CHECKED = true>
__device__ __forceinline__ float inv_sqrt_spec_nb(float x, bool& bad) {
  const float y = __builtin_amdgcn_rsqf(x);
  const float s0 = x * y;
  const float h = 0.5f * y;
  const float r = fmaf(-s0, s0, x);
  const float s1 = fmaf(r, h, s0);
  if constexpr (CHECKED) bad = bad | !in_fast_range(x);
  const float e = fmaf(-s1, y, 1.0f);
  const float inv = fmaf(e, y, y);
  return __uint_as_float(__float_as_uint(inv) + (((__float_as_uint(s1) & 0x7FFFFFu) == 0x7FFFFFu) ? 1u : 0u));
}

template <bool CHECKED = true>
__device__ __forceinline__ F3 normalize_nb(F3 v, bool& bad) { return v * inv_sqrt_spec_nb<CHECKED>(dot(v, v), bad); }

// normalize() of a vector that is ALREADY of unit length up to rounding (the second normalisation of the shading normal,
// the cross product of two orthogonal unit vectors, the cosine-weighted combination: pathtrace.cu:127,129,180).  Its squared
// length is within a few ulps of 1, so 1.0f / sqrtf(x) takes one of a handful of values: they are tabulated once per workgroup
// with the LITERAL expression (kUnitTab entries for the bit patterns 0x3F800000 - 16 ... + 16, SceneLds::inv1) and looked up
// by the integer distance of x's bits from 1.0f -- an LDS read instead of v_rsq_f32 and its correction steps.  Anything
// farther from 1 raises the redo flag like every other unverified input.
constexpr int kUnitTabHalf = 16, kUnitTabSize = 64;
__device__ __forceinline__ F3 normalize_unit_nb(F3 v, const float* __restrict__ tab, bool& bad) {
  const float x = dot(v, v);
  const uint32_t k = __float_as_uint(x) - (0x3F800000u - (uint32_t)kUnitTabHalf);
  bad = bad | (k > 2u * (uint32_t)kUnitTabHalf);
  return v * tab[k & (uint32_t)(kUnitTabSize - 1)];
}

__device__ __forceinline__ double sqrt_cr_nb(double x, bool& bad) {
  const uint32_t hi = (uint32_t)__double2hiint(x);
  bad = bad | ((hi - (423u << 20)) >= (1200u << 20));  // sign set, exponent outside 2^-600..2^600, inf, NaN
  double y = __builtin_amdgcn_rsq(x);
  double g = x * y;
  double h = y * 0.5;
  double r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g);
  h = __builtin_fma(h, r, h);
  double d = __builtin_fma(-g, g, x);
  g = __builtin_fma(d, h, g);
  d = __builtin_fma(-g, g, x);
  g = __builtin_fma(d, h, g);
  return g;
}

__device__ __forceinline__ float quotient_to_float_nb(double num, const RayConst& rc, bool& bad) {
  double q = num * rc.rden;
  double rem = __builtin_fma(-q, rc.den, num);
  q = __builtin_fma(rem, rc.rden, q);
  // Near a float rounding boundary the Markstein quotient's last ulp decides: redo.  The companion test of the literal
  // form -- q outside the float normal range -- is left to the callers: every one of them accepts the result only if
  // FLT_MIN <= t < 1e6 (`good`), and sends the lane to the literal loop otherwise.
  const uint32_t lo = (uint32_t)__double2loint(q);
  bad = bad | (((lo & 0x1FFFFFFFu) - 0x0FFFFFF8u) <= 0x10u);
  return (float)q;
}
constexpr float kMinGoodT = 1.17549435e-38f;  // FLT_MIN: a smaller positive t (a denormal float) is left to the literal code

// intersect_sphere for a sphere already known to satisfy det >= 0 is NOT assumed: the float part
// is recomputed and `real` returned exactly as the literal test would.
__device__ __forceinline__ bool intersect_sphere_nb_oc(F3 off, float c, F3 d, const RayConst& rc, float& t, bool& bad);
__device__ __forceinline__ bool intersect_sphere_nb(F3 o, F3 d, const RayConst& rc, float4 g, float& t, bool& bad) {
  const F3 off = mk3(o.x - g.x, o.y - g.y, o.z - g.z);
  return intersect_sphere_nb_oc(off, dot(off, off) - g.w, d, rc, t, bad);
}
// the same with off = o - centre and c = dot(off, off) - r*r supplied (primary rays: staged once per workgroup)
__device__ __forceinline__ bool intersect_sphere_nb_oc(F3 off, float c, F3 d, const RayConst& rc, float& t, bool& bad) {
  float b = 2.0f * dot(d, off);
  float bb = b * b;
  float det = bb - rc.a4 * c;
  double disc = (double)bb - rc.a4d * (double)c;
  bool bad_here = false;
  double sq = sqrt_cr_nb(disc, bad_here);
  double nb = (double)(-b);
  // The reference returns tNear if it is positive, else tFar (tNear <= tFar always: same
  // denominator 2a > 0, monotonic rounding).  tNear > 0 <=> nb - sq > 0 (the quotient keeps the
  // sign; a positive quotient that would round to float zero is outside the checked range and
  // goes to the literal code), so only that one quotient is evaluated.
  // = (n_near > 0.0) ? n_near : nb + sq with n_near = nb - sq: the difference is positive exactly when nb > sq, so the
  // sign of sq is chosen first (one select on the high word) and ONE sum formed
  const uint32_t sq_hi = (uint32_t)__double2hiint(sq);
  const double num = nb + __hiloint2double((int)((nb > sq) ? (sq_hi ^ 0x80000000u) : sq_hi), __double2loint(sq));
  t = quotient_to_float_nb(num, rc, bad_here);
  const bool real = det >= 0.0f;
  bad = bad | (real & bad_here);  // a negative or non-finite disc under det >= 0 lands here too
  return real;
}

// (float)sqrt(1.0 - (double)(ry * ry)) of getCosineWeightedNormal (:134) without FP64.  x = 1 - ry*ry is held exactly as
// hi + lo (Fast2Sum), sqrt(hi) ~ s0 comes from v_rsq_f32, and one Newton step with the FULL residual (x - s0*s0, lo included)
// lands within 2^-21 ulp of sqrt(x) before its final rounding.  That rounding is taken twice, with the step moved down and
// up by 2^-43 s0 (>= 2^-20 ulp: more than the step's own error, and far more than the 2^-30 ulp by which the reference's
// intermediate FP64 rounding can move a value across a tie): when both agree, the result is the reference's; when they do
// not, or when hi is zero or too small to matter (ry*ry > 1 - 2^-24), `bad` is raised and the caller redoes the step literally.
// Not argued but CHECKED for every float ry in [0, 1]: tests/test_unary_exhaustive_gpu.py.
__device__ __forceinline__ float oneminus_f32_nb(float ry, bool& bad) {
  const float w = ry * ry;
  const float hi = 1.0f - w;
  const float lo = (1.0f - hi) - w;  // exact: |1| >= |w|
  const float y = __builtin_amdgcn_rsqf(hi);
  const float s0 = hi * y;
  const float h = 0.5f * y;
  const float r = fmaf(-s0, s0, hi) + lo;
  const float delta = r * h;
  const float c = s0 * 1.1368684e-13f;  // 2^-43 s0
  const float sa = s0 + (delta - c), sb = s0 + (delta + c);
  bad = bad | (sa != sb) | !(hi > 5.9604645e-08f);
  return sa;
}

// the geometric part of one bounce: src/pathtrace.cu:163-166,178-180
struct BounceGeom {
  F3 normal;  // flipped shading normal
  F3 o, d;    // next ray
};

// TAB: unit_tab = SceneLds::inv1, the LDS table for normalize_unit_nb (scenes that are not staged have none: the general sequence)
template <bool FAST, bool TAB = false>
__device__ __forceinline__ BounceGeom bounce_geometry(F3 o, F3 d, float t, F3 centre, float u_az, float u_el, bool& bad,
                                                      const float* unit_tab = nullptr, uint32_t absmask = 0x7FFFFFFFu) {
  BounceGeom out;
  F3 pos = o + d * t;
  F3 normal = pos - centre;
  if constexpr (FAST) normal = normalize_nb(normal, bad); else normal = normalize(normal);
  if (!(dot(normal, d) < 0.0f)) normal = normal * -1.0f;
  out.normal = normal;
  out.o = pos + normal * 0.05f;
  // getCosineWeightedNormal, :126-136
  F3 dir, o1, o2;
  float ry, oneminus;
  if constexpr (FAST) {
    if constexpr (TAB) dir = normalize_unit_nb(normal, unit_tab, bad); else dir = normalize_nb(normal, bad);
    // Two range tests the table form makes redundant: unless `bad` is already raised, dir has unit length to within the
    // table's window, and ortho_vector drops the smaller of two of its components -- its squared length is within
    // [0.49, 1.01]; u_el is a curand_uniform value, 2^-33 .. 1 (Rng::bounce; the callers' placeholder is 0.5).
    o1 = normalize_nb<!TAB>(ortho_vector(dir), bad);
    if constexpr (TAB) o2 = normalize_unit_nb(cross(dir, o1), unit_tab, bad); else o2 = normalize_nb(cross(dir, o1), bad);
    ry = sqrt_cr_f32_nb<!TAB>(u_el, bad);
    if constexpr (TAB) oneminus = oneminus_f32_nb(ry, bad); else oneminus = (float)sqrt_cr_nb(1.0 - (double)(ry * ry), bad);
  } else {
    dir = normalize(normal);
    o1 = normalize(ortho_vector(dir));
    o2 = normalize(cross(dir, o1));
    ry = sqrtf(u_el);
    oneminus = (float)sqrt(1.0 - (double)(ry * ry));
  }
  // :131 u * 2.0f * 3.141592654f: doubling is exact, so the one product with 2 pi rounds the same real number
  float rx = FAST ? u_az * (2.0f * 3.141592654f) : u_az * 2.0f * 3.141592654f;
  float sn, cs;
  pt_sincos(rx, sn, cs, absmask);
  F3 a = o1 * (cs * oneminus);
  F3 b = o2 * (sn * oneminus);
  F3 c = dir * ry;
  F3 nd = (a + b) + c;
  if constexpr (FAST && TAB) out.d = normalize_unit_nb(nd, unit_tab, bad);
  else if constexpr (FAST) out.d = normalize_nb(nd, bad);
  else out.d = normalize(nd);
  return out;
}

// getCosineWeightedNormal with the cheap-but-identical sqrt / 1/sqrt sequences (variants >= 4)
__device__ __forceinline__ F3 cosine_weighted_fast(F3 dir, float u_az, float u_el) {
  dir = normalize_fast(dir);
  F3 o1 = normalize_fast(ortho_vector(dir));
  F3 o2 = normalize_fast(cross(dir, o1));
  float rx = u_az * 2.0f * 3.141592654f;
  float ry = sqrt_cr_f32(u_el);
  float oneminus = (float)sqrt_cr(1.0 - (double)(ry * ry));
  float sn, cs;
  pt_sincos(rx, sn, cs);
  F3 a = o1 * (cs * oneminus);
  F3 b = o2 * (sn * oneminus);
  F3 c = dir * ry;
  return (a + b) + c;
}

// denoise_kernel (src/denoise.cu:9-29, behind Denoiser::Denoise, main.cu:175) fused into the frame's epilogue: the pixel's display
// vertex (col, width - row, colour clamped to [0, 1] and packed as uchar4 {r, g, b, 1} into one float), from the very floats the
// frame store writes -- what pt_display_pack computes from the stored frame (pt_display.hip), without the second launch and the
// 12 B per pixel it reads back.
__device__ __forceinline__ void store_display_vertex(float* __restrict__ v, int width, int row, int col, float r, float g, float b) {
  const float c[3] = {r, g, b};
  uint32_t packed = 1u << 24;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const float x = fminf(fmaxf(c[k], 0.0f), 1.0f);                         // denoise.cu:18-20
    packed |= (uint32_t)(unsigned char)((double)x * 255.0) << (8 * k);      // :23
  }
  v[0] = (float)col;            // :26
  v[1] = (float)(width - row);  // :27
  v[2] = __uint_as_float(packed);  // :28
}

}  // namespace pt
