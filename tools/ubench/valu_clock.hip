// valu_clock.hip -- what does one SIMD really issue per cycle?  (tools/ubench/valu_cost.hip timed 0.3 ms kernels with
// events and ASSUMED that a grid of 8 blocks per CU puts exactly 8 waves on every SIMD: "1.125 ns per v_add_f32 per SIMD".)
// Here every wave records: s_memtime ticks (shader clock) and the constant 100 MHz real-time counter at its start and
// end, and WHERE it ran (HW_ID: SE / SH / CU / SIMD, XCC_ID).  The host groups the waves by SIMD, so the number of waves
// that really shared a SIMD is known instead of assumed, and reports per instruction:
//   - how the dispatcher distributed the waves (histogram of waves per SIMD),
//   - cycles per instruction per SIMD = (a wave's ticks / its instructions) / (waves on its SIMD), over SIMDs whose waves
//     all ran concurrently, for each occupancy that occurred,
//   - the clock the waves ran at (ticks per real-time ns), and the event-timed kernel duration for comparison.
// Kernels run > 50 ms (ITER is a run-time argument).
// Build: hipcc --offload-arch=gfx950 -O2 -o valu_clock valu_clock.hip ; run: ./valu_clock
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <map>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

struct Stamp { unsigned long long cyc, w0, w1; unsigned hwid, xcc; };

#define PROLOGUE                                                                                         \
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), w0 = wall_clock64();
#define EPILOGUE                                                                                         \
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), w1 = wall_clock64();                     \
    if ((threadIdx.x & 63) == 0)                                                                         \
      st[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] =                                          \
          Stamp{c1 - c0, w0, w1, (unsigned)__builtin_amdgcn_s_getreg(4 | (31 << 11)), (unsigned)__builtin_amdgcn_s_getreg(20 | (3 << 11))};

#define GROUP(ASM) ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)
#define KERNEL32(NAME, ASM)                                                                              \
  __global__ void __launch_bounds__(256) k_##NAME(float* out, Stamp* st, float seed, int iters) {        \
    float r0 = seed, r1 = seed + 1, r2 = seed + 2, r3 = seed + 3, r4 = seed + 4, r5 = seed + 5, r6 = seed + 6, \
          r7 = seed + 7, x = seed * 0.5f + 1.0f, y = 1.0001f;                                             \
    PROLOGUE                                                                                             \
    for (int i = 0; i < iters; i++) {                                                                    \
      asm volatile(GROUP(ASM) GROUP(ASM) GROUP(ASM) GROUP(ASM) GROUP(ASM) GROUP(ASM) GROUP(ASM) GROUP(ASM)  \
                   : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)      \
                   : "v"(x), "v"(y) : "vcc", "s20", "s21");                                              \
    }                                                                                                    \
    EPILOGUE                                                                                             \
    out[blockIdx.x * 256 + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;                         \
  }
#define KERNEL64(NAME, ASM)                                                                              \
  __global__ void __launch_bounds__(256) k_##NAME(float* out, Stamp* st, float seed, int iters) {        \
    double r0 = seed, r1 = seed + 1, r2 = seed + 2, r3 = seed + 3, r4 = seed + 4, r5 = seed + 5, r6 = seed + 6, \
           r7 = seed + 7, x = seed * 0.5 + 1.0, y = 1.0001;                                               \
    PROLOGUE                                                                                             \
    for (int i = 0; i < iters; i++) {                                                                    \
      asm volatile(GROUP(ASM) GROUP(ASM) GROUP(ASM) GROUP(ASM) GROUP(ASM) GROUP(ASM) GROUP(ASM) GROUP(ASM)  \
                   : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)      \
                   : "v"(x), "v"(y) : "vcc", "s20", "s21");                                              \
    }                                                                                                    \
    EPILOGUE                                                                                             \
    out[blockIdx.x * 256 + threadIdx.x] = (float)(r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7);                 \
  }

#define A_ADD32(n) "v_add_f32 %" #n ", %" #n ", %8\n"
#define A_MUL32(n) "v_mul_f32 %" #n ", %" #n ", %9\n"
#define A_FMA32(n) "v_fma_f32 %" #n ", %" #n ", %9, %8\n"
#define A_MIN32(n) "v_min_f32 %" #n ", %" #n ", %8\n"
#define A_MOV(n) "v_mov_b32 %" #n ", %8\n"
#define A_XOR(n) "v_xor_b32 %" #n ", %" #n ", %8\n"
#define A_RCP32(n) "v_rcp_f32 %" #n ", %" #n "\n"
#define A_CMPCND(n) "v_cmp_lt_f32 vcc, %" #n ", %8\nv_cndmask_b32 %" #n ", %" #n ", %9, vcc\n"
#define A_ADD64(n) "v_add_f64 %" #n ", %" #n ", %8\n"
#define A_FMA64(n) "v_fma_f64 %" #n ", %" #n ", %9, %8\n"
#define A_RCP64(n) "v_rcp_f64 %" #n ", %" #n "\n"
// round 3: the bit-field / three-operand integer forms the sphere screen is made of (is any of them full rate?)
#define A_BITOP3(n) "v_bitop3_b32 %" #n ", %" #n ", %8, %9 bitop3:0x6c\n"
#define A_ANDOR(n) "v_and_or_b32 %" #n ", %" #n ", %8, %9\n"
#define A_ANDOR_INL(n) "v_and_or_b32 %" #n ", %" #n ", -16, 3\n"
#define A_BFI(n) "v_bfi_b32 %" #n ", %8, %" #n ", %9\n"
#define A_MED3F(n) "v_med3_f32 %" #n ", %" #n ", %8, %9\n"
#define A_MED3U(n) "v_med3_u32 %" #n ", %" #n ", %8, %9\n"
#define A_MIN3U(n) "v_min3_u32 %" #n ", %" #n ", %8, %9\n"
#define A_MINU(n) "v_min_u32 %" #n ", %" #n ", %8\n"
#define A_MAXF(n) "v_max_f32 %" #n ", %" #n ", %8\n"
#define A_AND(n) "v_and_b32 %" #n ", %" #n ", %8\n"
#define A_OR3(n) "v_or3_b32 %" #n ", %" #n ", %8, %9\n"
#define A_ADD3(n) "v_add3_u32 %" #n ", %" #n ", %8, %9\n"
#define A_LSHLADD(n) "v_lshl_add_u32 %" #n ", %" #n ", 4, %9\n"
#define A_XAD(n) "v_xad_u32 %" #n ", %" #n ", %8, %9\n"
#define A_ADDABS(n) "v_add_f32_e64 %" #n ", |%" #n "|, %8\n"
#define A_MULNEG(n) "v_mul_f32_e64 %" #n ", %" #n ", -%9\n"
#define A_FMAMK(n) "v_fmamk_f32 %" #n ", %" #n ", 0x34000000, %8\n"
#define A_CMPS(n) "v_cmp_ngt_f32_e64 s[20:21], |%" #n "|, %8\n"
#define A_CMPV(n) "v_cmp_lt_f32 vcc, %" #n ", %8\n"
#define A_CNDS(n) "v_cndmask_b32_e64 %" #n ", %" #n ", %9, s[20:21]\n"
#define A_LSHL(n) "v_lshlrev_b32 %" #n ", 1, %" #n "\n"
#define A_SQRT32(n) "v_sqrt_f32 %" #n ", %" #n "\n"
#define A_MULLIT(n) "v_mul_f32 %" #n ", 0x7e800000, %" #n "\n"
#define A_PKMUL(n) "v_pk_mul_f32 %" #n ", %" #n ", %8\n"
#define A_PKADD(n) "v_pk_add_f32 %" #n ", %" #n ", %8\n"
#define A_SUBU(n) "v_sub_u32 %" #n ", %" #n ", %8\n"
#define A_MADU24(n) "v_mad_u32_u24 %" #n ", %" #n ", 4, %9\n"
#define A_MULU24(n) "v_mul_u32_u24 %" #n ", 16, %" #n "\n"
#define A_MADI24(n) "v_mad_i32_i24 %" #n ", %" #n ", %8, %9\n"
#define A_ALIGNBIT(n) "v_alignbit_b32 %" #n ", %" #n ", %8, 28\n"
#define A_BFE(n) "v_bfe_u32 %" #n ", %" #n ", 2, 30\n"
#define A_PERM(n) "v_perm_b32 %" #n ", %" #n ", %8, %9\n"
#define A_CVTF32U32(n) "v_cvt_f32_u32 %" #n ", %" #n "\n"
#define A_LDEXP(n) "v_ldexp_f32 %" #n ", %" #n ", %9\n"
#define A_CNDVCC(n) "v_cndmask_b32 %" #n ", %" #n ", %9, vcc\n"
#define A_FMA3(n) "v_fma_f32 %" #n ", %8, %9, %" #n "\n"
#define A_FMAC(n) "v_fmac_f32 %" #n ", %8, %9\n"
#define A_LSHLOR(n) "v_lshl_or_b32 %" #n ", %" #n ", 4, %9\n"
#define A_ADDLSHL(n) "v_add_lshl_u32 %" #n ", %" #n ", %8, 2\n"
#define A_LSHR(n) "v_lshrrev_b32 %" #n ", 2, %" #n "\n"
#define A_ASHR(n) "v_ashrrev_i32 %" #n ", 31, %" #n "\n"
#define A_MINI(n) "v_min_i32 %" #n ", %" #n ", %8\n"
#define A_SUBF32SDWA(n) "v_add_f32_sdwa %" #n ", %" #n ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n"
#define A_MOVDPP(n) "v_mov_b32_dpp %" #n ", %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define A_MULF64(n) "v_mul_f64 %" #n ", %" #n ", %9\n"
#define A_CVTF64F32(n) "v_cvt_f64_f32 %" #n ", %8\n"
// one compare feeding three selects (the shape of "flip the normal", "orthogonal vector", the sincos quadrant), through VCC and
// through an SGPR pair; and selects alone on a VCC written once before the loop (cndmask_vcc above: 22.9 cycles each!)
#define A_CMP3CND_VCC(n) "v_cmp_lt_f32 vcc, %" #n ", %8\nv_cndmask_b32 %" #n ", %" #n ", %9, vcc\nv_cndmask_b32 %" #n ", %" #n ", %8, vcc\nv_cndmask_b32 %" #n ", %" #n ", %9, vcc\n"
#define A_CMP3CND_SGPR(n) "v_cmp_lt_f32_e64 s[20:21], %" #n ", %8\nv_cndmask_b32_e64 %" #n ", %" #n ", %9, s[20:21]\nv_cndmask_b32_e64 %" #n ", %" #n ", %8, s[20:21]\nv_cndmask_b32_e64 %" #n ", %" #n ", %9, s[20:21]\n"
#define A_CMP3XOR(n) "v_cmp_lt_f32_e64 s[20:21], %" #n ", %8\nv_cndmask_b32_e64 %" #n ", 0, %9, s[20:21]\nv_xor_b32 %" #n ", %" #n ", %8\nv_xor_b32 %" #n ", %" #n ", %9\n"
KERNEL32(cmp_3cnd_vcc, A_CMP3CND_VCC)
KERNEL32(cmp_3cnd_sgpr, A_CMP3CND_SGPR)
KERNEL32(cmp_cnd_2xor, A_CMP3XOR)
KERNEL32(mad_u32_u24, A_MADU24)
KERNEL32(mul_u32_u24, A_MULU24)
KERNEL32(mad_i32_i24, A_MADI24)
KERNEL32(alignbit_b32, A_ALIGNBIT)
KERNEL32(bfe_u32, A_BFE)
KERNEL32(perm_b32, A_PERM)
KERNEL32(cvt_f32_u32, A_CVTF32U32)
KERNEL32(ldexp_f32, A_LDEXP)
KERNEL32(cndmask_vcc, A_CNDVCC)
KERNEL32(fma_f32_3src, A_FMA3)
KERNEL32(fmac_f32, A_FMAC)
KERNEL32(lshl_or_b32, A_LSHLOR)
KERNEL32(add_lshl_u32, A_ADDLSHL)
KERNEL32(lshrrev_b32, A_LSHR)
KERNEL32(ashrrev_i32, A_ASHR)
KERNEL32(min_i32, A_MINI)
KERNEL32(add_f32_sdwa, A_SUBF32SDWA)
KERNEL32(mov_b32_dpp, A_MOVDPP)
KERNEL64(mul_f64, A_MULF64)
KERNEL32(bitop3_b32, A_BITOP3)
KERNEL32(and_or_b32, A_ANDOR)
KERNEL32(and_or_inline, A_ANDOR_INL)
KERNEL32(bfi_b32, A_BFI)
KERNEL32(med3_f32, A_MED3F)
KERNEL32(med3_u32, A_MED3U)
KERNEL32(min3_u32, A_MIN3U)
KERNEL32(min_u32, A_MINU)
KERNEL32(max_f32, A_MAXF)
KERNEL32(and_b32, A_AND)
KERNEL32(or3_b32, A_OR3)
KERNEL32(add3_u32, A_ADD3)
KERNEL32(lshl_add_u32, A_LSHLADD)
KERNEL32(xad_u32, A_XAD)
KERNEL32(add_f32_abs, A_ADDABS)
KERNEL32(mul_f32_neg, A_MULNEG)
KERNEL32(fmamk_f32, A_FMAMK)
KERNEL32(cmp_to_sgpr, A_CMPS)
KERNEL32(cmp_to_vcc, A_CMPV)
KERNEL32(cndmask_sgpr, A_CNDS)
KERNEL32(lshlrev_b32, A_LSHL)
KERNEL32(sqrt_f32, A_SQRT32)
KERNEL32(mul_f32_lit, A_MULLIT)
KERNEL64(pk_mul_f32, A_PKMUL)
KERNEL64(pk_add_f32, A_PKADD)
KERNEL32(sub_u32, A_SUBU)
KERNEL32(add_f32, A_ADD32)
KERNEL32(mul_f32, A_MUL32)
KERNEL32(fma_f32, A_FMA32)
KERNEL32(min_f32, A_MIN32)
KERNEL32(mov_b32, A_MOV)
KERNEL32(xor_b32, A_XOR)
KERNEL32(rcp_f32, A_RCP32)
KERNEL32(cmp_cnd_pair, A_CMPCND)
KERNEL64(add_f64, A_ADD64)
KERNEL64(fma_f64, A_FMA64)
KERNEL64(rcp_f64, A_RCP64)

typedef void (*kfn)(float*, Stamp*, float, int);
struct Entry { const char* name; kfn fn; int per_group; };  // instructions per ASM group
#define E(n, k) {#n, k_##n, k}

int main(int argc, char** argv) {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const int iters = argc > 1 ? atoi(argv[1]) : 65536;  // x 64 instructions
  float* out;
  Stamp* st;
  const int max_blocks = cus * 8;
  CHECK(hipMalloc(&out, (size_t)max_blocks * 256 * sizeof(float)));
  CHECK(hipMalloc(&st, (size_t)max_blocks * 4 * sizeof(Stamp)));
  std::vector<Stamp> hst((size_t)max_blocks * 4);
  Entry es[] = {E(add_f32, 1), E(mul_f32, 1), E(fma_f32, 1), E(min_f32, 1), E(mov_b32, 1), E(xor_b32, 1), E(rcp_f32, 1),
                E(cmp_cnd_pair, 2), E(add_f64, 1), E(fma_f64, 1), E(rcp_f64, 1)};
  Entry ext[] = {E(add_f32, 1), E(bitop3_b32, 1), E(and_or_b32, 1), E(and_or_inline, 1), E(bfi_b32, 1), E(med3_f32, 1), E(med3_u32, 1),
                 E(min3_u32, 1), E(min_u32, 1), E(max_f32, 1), E(and_b32, 1), E(or3_b32, 1), E(add3_u32, 1), E(lshl_add_u32, 1),
                 E(xad_u32, 1), E(add_f32_abs, 1), E(mul_f32_neg, 1), E(fmamk_f32, 1), E(cmp_to_sgpr, 1), E(cmp_to_vcc, 1),
                 E(cndmask_sgpr, 1), E(lshlrev_b32, 1), E(sqrt_f32, 1), E(mul_f32_lit, 1), E(pk_mul_f32, 1), E(pk_add_f32, 1), E(sub_u32, 1)};
  Entry ext2[] = {E(add_f32, 1), E(mad_u32_u24, 1), E(mul_u32_u24, 1), E(mad_i32_i24, 1), E(alignbit_b32, 1), E(bfe_u32, 1), E(perm_b32, 1),
                  E(cvt_f32_u32, 1), E(ldexp_f32, 1), E(cndmask_vcc, 1), E(fma_f32_3src, 1), E(fmac_f32, 1), E(lshl_or_b32, 1),
                  E(add_lshl_u32, 1), E(lshrrev_b32, 1), E(ashrrev_i32, 1), E(min_i32, 1), E(add_f32_sdwa, 1), E(mov_b32_dpp, 1), E(mul_f64, 1)};
  Entry ext3[] = {E(add_f32, 1), E(cmp_3cnd_vcc, 4), E(cmp_3cnd_sgpr, 4), E(cmp_cnd_2xor, 4), E(cndmask_sgpr, 1), E(cndmask_vcc, 1), E(lshlrev_b32, 1), E(lshrrev_b32, 1)};
  const bool extended = argc > 2 && (argv[2][0] == 'x' || argv[2][0] == 'y' || argv[2][0] == 'z');  // x, y: the two round-3 lists  // ./valu_clock 65536 x : the round-3 list, 8 and 4 waves per SIMD only
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  printf("%s, %d CUs, clockRate %d kHz, %d iterations x 64 instructions per wave\n", prop.name, cus, prop.clockRate, iters);
  std::vector<Entry> run;
  if (extended && argv[2][0] == 'z') run.assign(std::begin(ext3), std::end(ext3)); else if (extended && argv[2][0] == 'y') run.assign(std::begin(ext2), std::end(ext2)); else if (extended) run.assign(std::begin(ext), std::end(ext)); else run.assign(std::begin(es), std::end(es));
  for (auto& e : run) {
    for (int wps : {8, 5, 4, 2, 1}) {
      if (extended && wps != 8 && wps != 4) continue;
      const int nb = cus * wps;  // wps blocks of 4 waves per CU: wps waves per SIMD IF the dispatcher spreads them evenly
      hipLaunchKernelGGL(e.fn, dim3(nb), dim3(256), 0, 0, out, st, 1.0f, 4096);  // warm
      CHECK(hipDeviceSynchronize());
      CHECK(hipEventRecord(a));
      hipLaunchKernelGGL(e.fn, dim3(nb), dim3(256), 0, 0, out, st, 1.0f, iters);
      CHECK(hipEventRecord(b));
      CHECK(hipEventSynchronize(b));
      float ms;
      CHECK(hipEventElapsedTime(&ms, a, b));
      const int nw = nb * 4;
      CHECK(hipMemcpy(hst.data(), st, (size_t)nw * sizeof(Stamp), hipMemcpyDeviceToHost));
      // group by SIMD: xcc | se | sh | cu | simd
      std::map<unsigned, std::vector<int>> simds;
      unsigned long long t_first = ~0ull, t_last = 0;
      for (int i = 0; i < nw; i++) {
        const unsigned h = hst[i].hwid;
        const unsigned key = ((hst[i].xcc & 15u) << 12) | (((h >> 13) & 7u) << 9) | (((h >> 12) & 1u) << 8) | (((h >> 8) & 15u) << 4) | ((h >> 4) & 3u);
        simds[key].push_back(i);
        t_first = std::min(t_first, hst[i].w0);
        t_last = std::max(t_last, hst[i].w1);
      }
      const double instr_wave = (double)iters * 64 * e.per_group;
      std::map<int, int> hist;
      for (auto& kv : simds) hist[(int)kv.second.size()]++;
      double cyc_sum = 0, ghz_sum = 0;
      for (int i = 0; i < nw; i++) cyc_sum += (double)hst[i].cyc;
      for (int i = 0; i < nw; i++) ghz_sum += (double)hst[i].cyc / ((double)(hst[i].w1 - hst[i].w0) * 10.0);
      const bool even = hist.size() == 1 && hist.begin()->first == wps;
      printf("%-13s %d waves/SIMD%s: event %8.3f ms = %6.3f ns/instr/SIMD; wave ticks -> %6.3f cycles/instr/SIMD at %.3f GHz (ticks per real-time ns)",
             e.name, wps, even ? "" : " (UNEVEN)", ms, ms * 1e6 / (instr_wave * wps), cyc_sum / nw / instr_wave / wps, ghz_sum / nw);
      if (!even) {
        printf("; waves/SIMD histogram:");
        for (auto& h : hist) printf(" %dx%d", h.first, h.second);
      }
      printf("\n");
    }
  }
  return 0;
}
