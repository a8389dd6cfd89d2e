// valu_cost.hip -- measures the issue cost (SIMD cycles per wave64 instruction) of the VALU
// instructions the path-trace kernel is made of, on the GPU it runs on.  8 waves per SIMD,
// 8 independent dependency chains per wave, so the number is throughput, not latency.
// Build: hipcc --offload-arch=gfx950 -O2 -o valu_cost valu_cost.hip ; run: ./valu_cost
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

#define REP8(op) op(0) op(1) op(2) op(3) op(4) op(5) op(6) op(7)
#define ITER 4096

#define KERNEL32(NAME, ASM)                                                              \
  __global__ void __launch_bounds__(256) k_##NAME(float* out, float seed) {              \
    float r0 = seed, r1 = seed + 1, r2 = seed + 2, r3 = seed + 3, r4 = seed + 4, r5 = seed + 5, r6 = seed + 6, \
          r7 = seed + 7, x = seed * 0.5f + 1.0f, y = 1.0001f;                             \
    for (int i = 0; i < ITER; i++) {                                                     \
      asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)               \
                   : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) \
                   : "v"(x), "v"(y));                                                    \
    }                                                                                    \
    out[blockIdx.x * 256 + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;         \
  }

#define KERNEL64(NAME, ASM)                                                              \
  __global__ void __launch_bounds__(256) k_##NAME(float* out, float seed) {              \
    double r0 = seed, r1 = seed + 1, r2 = seed + 2, r3 = seed + 3, r4 = seed + 4, r5 = seed + 5, r6 = seed + 6, \
           r7 = seed + 7, x = seed * 0.5 + 1.0, y = 1.0001;                               \
    for (int i = 0; i < ITER; i++) {                                                     \
      asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)               \
                   : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) \
                   : "v"(x), "v"(y));                                                    \
    }                                                                                    \
    out[blockIdx.x * 256 + threadIdx.x] = (float)(r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7); \
  }

#define A_ADD32(n) "v_add_f32 %" #n ", %" #n ", %8\n"
#define A_MUL32(n) "v_mul_f32 %" #n ", %" #n ", %9\n"
#define A_FMA32(n) "v_fma_f32 %" #n ", %" #n ", %9, %8\n"
#define A_MAX32(n) "v_max_f32 %" #n ", %" #n ", %8\n"
#define A_RCP32(n) "v_rcp_f32 %" #n ", %" #n "\n"
#define A_SQRT32(n) "v_sqrt_f32 %" #n ", %" #n "\n"
#define A_RSQ32(n) "v_rsq_f32 %" #n ", %" #n "\n"
#define A_CNDMASK(n) "v_cndmask_b32 %" #n ", %" #n ", %8, vcc\n"
#define A_CMP(n) "v_cmp_lt_f32 vcc, %" #n ", %8\n"
#define A_CMPS(n) "v_cmp_lt_f32 s[20:21], %" #n ", %8\n"
#define A_XOR(n) "v_xor_b32 %" #n ", %" #n ", %8\n"
#define A_ADDU(n) "v_add_u32 %" #n ", %" #n ", %8\n"
#define A_LSHL(n) "v_lshlrev_b32 %" #n ", 3, %" #n "\n"
#define A_BFI(n) "v_bfi_b32 %" #n ", %8, %" #n ", %9\n"
#define A_MULLO(n) "v_mul_lo_u32 %" #n ", %" #n ", %8\n"
#define A_MULHI(n) "v_mul_hi_u32 %" #n ", %" #n ", %8\n"
#define A_CVTU(n) "v_cvt_f32_u32 %" #n ", %" #n "\n"
#define A_DIVSCALE32(n) "v_div_scale_f32 %" #n ", vcc, %" #n ", %8, %9\n"
#define A_DIVFMAS32(n) "v_div_fmas_f32 %" #n ", %" #n ", %8, %9\n"
#define A_DIVFIXUP32(n) "v_div_fixup_f32 %" #n ", %" #n ", %8, %9\n"
#define A_MOV(n) "v_mov_b32 %" #n ", %8\n"

#define A_ADD64(n) "v_add_f64 %" #n ", %" #n ", %8\n"
#define A_MUL64(n) "v_mul_f64 %" #n ", %" #n ", %9\n"
#define A_FMA64(n) "v_fma_f64 %" #n ", %" #n ", %9, %8\n"
#define A_RCP64(n) "v_rcp_f64 %" #n ", %" #n "\n"
#define A_RSQ64(n) "v_rsq_f64 %" #n ", %" #n "\n"
#define A_SQRT64(n) "v_sqrt_f64 %" #n ", %" #n "\n"
#define A_LDEXP64(n) "v_ldexp_f64 %" #n ", %" #n ", 1\n"
#define A_CMP64(n) "v_cmp_lt_f64 vcc, %" #n ", %8\n"
#define A_DIVSCALE64(n) "v_div_scale_f64 %" #n ", vcc, %" #n ", %8, %9\n"
#define A_DIVFMAS64(n) "v_div_fmas_f64 %" #n ", %" #n ", %8, %9\n"
#define A_DIVFIXUP64(n) "v_div_fixup_f64 %" #n ", %" #n ", %8, %9\n"
#define A_PKADD(n) "v_pk_add_f32 %" #n ", %" #n ", %8\n"
#define A_PKMUL(n) "v_pk_mul_f32 %" #n ", %" #n ", %9\n"
#define A_PKFMA(n) "v_pk_fma_f32 %" #n ", %" #n ", %9, %8\n"

#define A_CMPCND(n) "v_cmp_lt_f32 vcc, %" #n ", %8\nv_cndmask_b32 %" #n ", %" #n ", %9, vcc\n"
#define A_CMPCND64(n) "v_cmp_lt_f32 s[20:21], %" #n ", %8\nv_cndmask_b32_e64 %" #n ", %" #n ", %9, s[20:21]\n"
#define A_CND64(n) "v_cndmask_b32_e64 %" #n ", %" #n ", %9, s[20:21]\n"
#define A_CNDVCC2(n) "v_cndmask_b32 %" #n ", %8, %9, vcc\n"
#define A_MIN32(n) "v_min_f32 %" #n ", %" #n ", %8\n"
#define A_MED3(n) "v_med3_f32 %" #n ", %" #n ", %8, %9\n"
#define A_MIN3(n) "v_min3_f32 %" #n ", %" #n ", %8, %9\n"
#define A_MINU(n) "v_min_u32 %" #n ", %" #n ", %8\n"
#define A_MED3U(n) "v_med3_u32 %" #n ", %" #n ", %8, %9\n"
#define A_ANDOR(n) "v_and_or_b32 %" #n ", %" #n ", %8, %9\n"
#define A_SUB32(n) "v_sub_f32 %" #n ", %" #n ", %8\n"
#define A_ADDMOD(n) "v_add_f32 %" #n ", -%" #n ", |%8|\n"
#define A_FMAC(n) "v_fmac_f32 %" #n ", %8, %9\n"
#define A_FMAAK(n) "v_fmaak_f32 %" #n ", %" #n ", %8, 0x3f800000\n"
#define A_MULLIT(n) "v_mul_f32 %" #n ", 0x40490fdb, %" #n "\n"
#define A_ADD3U(n) "v_add3_u32 %" #n ", %" #n ", %8, %9\n"
#define A_RNDNE(n) "v_rndne_f32 %" #n ", %" #n "\n"
#define A_CVTI(n) "v_cvt_i32_f32 %" #n ", %" #n "\n"
#define A_LDEXP32(n) "v_ldexp_f32 %" #n ", %" #n ", 1\n"
#define A_CMPCLASS(n) "v_cmp_class_f32 vcc, %" #n ", %8\n"
KERNEL32(cmp_cnd_vcc_pair, A_CMPCND)
KERNEL32(cmp_cnd_sgpr_pair, A_CMPCND64)
KERNEL32(cndmask_e64_sgpr, A_CND64)
KERNEL32(cndmask_vcc_nodep, A_CNDVCC2)
KERNEL32(min_f32, A_MIN32)
KERNEL32(med3_f32, A_MED3)
KERNEL32(min3_f32, A_MIN3)
KERNEL32(min_u32, A_MINU)
KERNEL32(med3_u32, A_MED3U)
KERNEL32(and_or_b32, A_ANDOR)
KERNEL32(sub_f32, A_SUB32)
KERNEL32(add_f32_mods, A_ADDMOD)
KERNEL32(fmac_f32, A_FMAC)
KERNEL32(fmaak_f32, A_FMAAK)
KERNEL32(mul_f32_literal, A_MULLIT)
KERNEL32(add3_u32, A_ADD3U)
KERNEL32(rndne_f32, A_RNDNE)
KERNEL32(cvt_i32_f32, A_CVTI)
KERNEL32(ldexp_f32, A_LDEXP32)
KERNEL32(cmp_class_f32, A_CMPCLASS)
KERNEL32(add_f32, A_ADD32)
KERNEL32(mul_f32, A_MUL32)
KERNEL32(fma_f32, A_FMA32)
KERNEL32(max_f32, A_MAX32)
KERNEL32(rcp_f32, A_RCP32)
KERNEL32(sqrt_f32, A_SQRT32)
KERNEL32(rsq_f32, A_RSQ32)
KERNEL32(cndmask, A_CNDMASK)
KERNEL32(cmp_vcc, A_CMP)
KERNEL32(cmp_sgpr, A_CMPS)
KERNEL32(xor_b32, A_XOR)
KERNEL32(add_u32, A_ADDU)
KERNEL32(lshl_b32, A_LSHL)
KERNEL32(bfi_b32, A_BFI)
KERNEL32(mul_lo_u32, A_MULLO)
KERNEL32(mul_hi_u32, A_MULHI)
KERNEL32(cvt_f32_u32, A_CVTU)
KERNEL32(div_scale_f32, A_DIVSCALE32)
KERNEL32(div_fmas_f32, A_DIVFMAS32)
KERNEL32(div_fixup_f32, A_DIVFIXUP32)
KERNEL32(mov_b32, A_MOV)
KERNEL64(add_f64, A_ADD64)
KERNEL64(mul_f64, A_MUL64)
KERNEL64(fma_f64, A_FMA64)
KERNEL64(rcp_f64, A_RCP64)
KERNEL64(rsq_f64, A_RSQ64)
KERNEL64(sqrt_f64, A_SQRT64)
KERNEL64(ldexp_f64, A_LDEXP64)
KERNEL64(cmp_f64, A_CMP64)
KERNEL64(div_scale_f64, A_DIVSCALE64)
KERNEL64(div_fmas_f64, A_DIVFMAS64)
KERNEL64(div_fixup_f64, A_DIVFIXUP64)
KERNEL64(pk_add_f32, A_PKADD)
KERNEL64(pk_mul_f32, A_PKMUL)
KERNEL64(pk_fma_f32, A_PKFMA)

// conversions need mixed register widths: separate kernels
__global__ void __launch_bounds__(256) k_cvt_f64_f32(float* out, float seed) {
  float s0 = seed, s1 = seed + 1, s2 = seed + 2, s3 = seed + 3;
  double d0 = 0, d1 = 0, d2 = 0, d3 = 0;
  for (int i = 0; i < ITER; i++) {
    asm volatile("v_cvt_f64_f32 %0, %4\nv_cvt_f64_f32 %1, %5\nv_cvt_f64_f32 %2, %6\nv_cvt_f64_f32 %3, %7\n"
                 "v_cvt_f64_f32 %0, %4\nv_cvt_f64_f32 %1, %5\nv_cvt_f64_f32 %2, %6\nv_cvt_f64_f32 %3, %7\n"
                 : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(s0), "v"(s1), "v"(s2), "v"(s3));
  }
  out[blockIdx.x * 256 + threadIdx.x] = (float)(d0 + d1 + d2 + d3);
}
__global__ void __launch_bounds__(256) k_cvt_f32_f64(float* out, float seed) {
  double d0 = seed, d1 = seed + 1, d2 = seed + 2, d3 = seed + 3;
  float s0 = 0, s1 = 0, s2 = 0, s3 = 0;
  for (int i = 0; i < ITER; i++) {
    asm volatile("v_cvt_f32_f64 %0, %4\nv_cvt_f32_f64 %1, %5\nv_cvt_f32_f64 %2, %6\nv_cvt_f32_f64 %3, %7\n"
                 "v_cvt_f32_f64 %0, %4\nv_cvt_f32_f64 %1, %5\nv_cvt_f32_f64 %2, %6\nv_cvt_f32_f64 %3, %7\n"
                 : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3) : "v"(d0), "v"(d1), "v"(d2), "v"(d3));
  }
  out[blockIdx.x * 256 + threadIdx.x] = s0 + s1 + s2 + s3;
}

typedef void (*kfn)(float*, float);
struct Entry { const char* name; kfn fn; };
#define E(n) {#n, k_##n}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount, blocks = cus * 8;  // 8 blocks x 4 waves = 8 waves per SIMD
  float* out;
  CHECK(hipMalloc(&out, (size_t)blocks * 256 * sizeof(float)));
  Entry es[] = {E(add_f32), E(sub_f32), E(add_f32_mods), E(fmac_f32), E(fmaak_f32), E(mul_f32_literal), E(min_f32), E(med3_f32), E(min3_f32), E(min_u32), E(med3_u32), E(and_or_b32), E(add3_u32), E(rndne_f32), E(cvt_i32_f32), E(ldexp_f32), E(cmp_class_f32), E(cmp_cnd_vcc_pair), E(cmp_cnd_sgpr_pair), E(cndmask_e64_sgpr), E(cndmask_vcc_nodep), E(mul_f32), E(fma_f32), E(max_f32), E(mov_b32), E(cndmask), E(cmp_vcc), E(cmp_sgpr), E(xor_b32),
                E(add_u32), E(lshl_b32), E(bfi_b32), E(mul_lo_u32), E(mul_hi_u32), E(cvt_f32_u32), E(rcp_f32), E(sqrt_f32),
                E(rsq_f32), E(div_scale_f32), E(div_fmas_f32), E(div_fixup_f32), E(pk_add_f32), E(pk_mul_f32), E(pk_fma_f32),
                E(add_f64), E(mul_f64), E(fma_f64), E(ldexp_f64), E(cmp_f64), E(cvt_f64_f32), E(cvt_f32_f64), E(rcp_f64),
                E(rsq_f64), E(sqrt_f64), E(div_scale_f64), E(div_fmas_f64), E(div_fixup_f64)};
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  double base = 0;
  printf("%s, %d CUs, clockRate %d kHz; 8 waves/SIMD, %d x 8 instr per wave\n", prop.name, cus, prop.clockRate, ITER);
  printf("%-16s %10s %14s %10s\n", "instruction", "ms", "ns/instr/SIMD", "vs add_f32");
  for (auto& e : es) {
    hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, out, 1.0f);
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
      CHECK(hipEventRecord(a));
      hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, out, 1.0f);
      CHECK(hipEventRecord(b));
      CHECK(hipEventSynchronize(b));
      float ms;
      CHECK(hipEventElapsedTime(&ms, a, b));
      if (ms < best) best = ms;
    }
    const double instr_per_simd = (double)ITER * 8 * 8;  // per wave x 8 waves per SIMD
    const double ns = best * 1e6 / instr_per_simd;
    if (base == 0) base = ns;
    printf("%-16s %10.3f %14.3f %10.2f\n", e.name, best, ns, ns / base);
  }
  // per-wave issue limit: same kernels with 1, 2, 4, 8 waves per SIMD (one block = 4 waves = 1 per SIMD)
  printf("\nissue rate vs occupancy (ns per instruction per WAVE; ideal = constant x waves)\n");
  Entry sweep[] = {E(add_f32), E(fma_f32), E(fma_f64), E(cmp_cnd_vcc_pair), E(rcp_f32)};
  for (auto& e : sweep) {
    printf("%-18s", e.name);
    for (int wps = 1; wps <= 8; wps *= 2) {
      const int nb = cus * wps;
      hipLaunchKernelGGL(e.fn, dim3(nb), dim3(256), 0, 0, out, 1.0f);
      CHECK(hipDeviceSynchronize());
      float best = 1e30f;
      for (int rep = 0; rep < 3; rep++) {
        CHECK(hipEventRecord(a));
        hipLaunchKernelGGL(e.fn, dim3(nb), dim3(256), 0, 0, out, 1.0f);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms;
        CHECK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
      }
      printf("  %d w/SIMD: %6.3f ns/instr/wave (%5.3f per SIMD)", wps, best * 1e6 / ((double)ITER * 8), best * 1e6 / ((double)ITER * 8 * wps));
    }
    printf("\n");
  }
  return 0;
}
