// valu_banks.hip -- do VGPR operand banks matter for the issue rate of three-source VALU instructions on gfx950?
// Each kernel runs 64 instructions per loop trip on HARD-CODED registers (destinations v20..v27 in turn, sources chosen by bank =
// register number mod 4), 8 waves per SIMD, timed with events.  Build: hipcc --offload-arch=gfx950 -O2 -o valu_banks valu_banks.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

#define CLOB "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "s20", "s22", "s23", "scc", "vcc"
#define INIT "v_mov_b32 v20, 1.0\nv_mov_b32 v21, 1.0\nv_mov_b32 v22, 1.0\nv_mov_b32 v23, 1.0\nv_mov_b32 v24, 1.0\nv_mov_b32 v25, 1.0\nv_mov_b32 v26, 1.0\nv_mov_b32 v27, 1.0\n" \
             "v_mov_b32 v28, 0.5\nv_mov_b32 v29, 0.5\nv_mov_b32 v30, 0.5\nv_mov_b32 v31, 0.5\nv_mov_b32 v32, 0.5\nv_mov_b32 v33, 0.5\nv_mov_b32 v34, 0.5\nv_mov_b32 v35, 0.5\n" \
             "v_mov_b32 v36, 0.5\nv_mov_b32 v37, 0.5\nv_mov_b32 v38, 0.5\nv_mov_b32 v39, 0.5\n"
// one group = 8 instructions, destination v20..v27
#define G8(OP, S) OP " v20, " S "\n" OP " v21, " S "\n" OP " v22, " S "\n" OP " v23, " S "\n" OP " v24, " S "\n" OP " v25, " S "\n" OP " v26, " S "\n" OP " v27, " S "\n"
#define BODY(OP, S) G8(OP, S) G8(OP, S) G8(OP, S) G8(OP, S) G8(OP, S) G8(OP, S) G8(OP, S) G8(OP, S)
#define KERNEL(NAME, OP, S)                                                                                   \
  __global__ void __launch_bounds__(256) k_##NAME(float* out, int iters) {                                    \
    float r;                                                                                                  \
    asm volatile(INIT "s_mov_b32 s20, %1\n1:\n" BODY(OP, S) "s_sub_u32 s20, s20, 1\ns_cmp_lg_u32 s20, 0\ns_cbranch_scc1 1b\n" \
                 "v_add_f32 %0, v20, v21\nv_add_f32 %0, %0, v22\nv_add_f32 %0, %0, v23\n"                    \
                 : "=v"(r) : "s"(iters) : CLOB);                                                              \
    out[blockIdx.x * 256 + threadIdx.x] = r;                                                                  \
  }
KERNEL(add_2banks, "v_add_f32", "v28, v29")
KERNEL(add_1bank, "v_add_f32", "v28, v32")
KERNEL(fma_3banks, "v_fma_f32", "v28, v29, v30")
KERNEL(fma_2banks, "v_fma_f32", "v28, v32, v30")
KERNEL(fma_1bank, "v_fma_f32", "v28, v32, v36")
KERNEL(fma_same_reg, "v_fma_f32", "v28, v28, v29")
KERNEL(fma_abs_neg, "v_fma_f32", "v28, |v29|, -v30")
KERNEL(fmac_2banks, "v_fmac_f32", "v28, v29")
KERNEL(fmac_1bank, "v_fmac_f32", "v28, v32")
KERNEL(bitop3_3banks, "v_bitop3_b32", "v28, v29, v30 bitop3:0xfe")
KERNEL(bitop3_1bank, "v_bitop3_b32", "v28, v32, v36 bitop3:0xfe")
KERNEL(bitop3_const, "v_bitop3_b32", "v28, 15, 3 bitop3:0xba")
KERNEL(minu_2banks, "v_min_u32", "v28, v29")
KERNEL(mul_2banks, "v_mul_f32", "v28, v29")
KERNEL(fmamk, "v_fmamk_f32", "v28, 0x34000000, v29")
KERNEL(sub_2banks, "v_sub_f32", "v28, v29")
// which pairs conflict?  src0 = v28 fixed; src1 swept with src2 = v39, then src2 swept with src1 = v29
KERNEL(s1_v29, "v_fma_f32", "v28, v29, v39")
KERNEL(s1_v30, "v_fma_f32", "v28, v30, v39")
KERNEL(s1_v31, "v_fma_f32", "v28, v31, v39")
KERNEL(s1_v32, "v_fma_f32", "v28, v32, v39")
KERNEL(s1_v33, "v_fma_f32", "v28, v33, v39")
KERNEL(s1_v36, "v_fma_f32", "v28, v36, v39")
KERNEL(s2_v30, "v_fma_f32", "v28, v29, v30")
KERNEL(s2_v32, "v_fma_f32", "v28, v29, v32")
KERNEL(s2_v33, "v_fma_f32", "v28, v29, v33")
KERNEL(s2_v36, "v_fma_f32", "v28, v29, v36")
KERNEL(s2_v37, "v_fma_f32", "v28, v29, v37")
KERNEL(s12_same, "v_fma_f32", "v28, v33, v37")
KERNEL(bitop3_sgpr, "v_bitop3_b32", "v28, v29, s4 bitop3:0xb1")
KERNEL(bitop3_sgpr_same, "v_bitop3_b32", "v28, v30, s4 bitop3:0xb1")
KERNEL(add_sgpr, "v_add_f32", "s4, v29")
KERNEL(fmamk_2, "v_fmamk_f32", "v28, 0x34000000, v30")
KERNEL(fmac_sgpr, "v_fmac_f32", "s4, v29")
KERNEL(andor_sgpr, "v_and_or_b32", "v28, s4, v29")
KERNEL(fma_3odd, "v_fma_f32", "v29, v31, v33")
KERNEL(fma_same_even, "v_fma_f32", "v28, v28, v30")
KERNEL(fma_sgpr_odd, "v_fma_f32", "v29, s4, v31")
KERNEL(fma_sgpr_mixed, "v_fma_f32", "v28, s4, v31")
KERNEL(fma_sgpr, "v_fma_f32", "v28, s4, v32")
KERNEL(fma_const, "v_fma_f32", "v28, 0.5, v32")

typedef void (*kfn)(float*, int);
struct Entry { const char* name; kfn fn; };
#define E(n) {#n, k_##n}
int main(int argc, char** argv) {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount, iters = argc > 1 ? atoi(argv[1]) : 65536;
  float* out;
  CHECK(hipMalloc(&out, (size_t)cus * 8 * 256 * sizeof(float)));
  Entry es[] = {E(add_2banks), E(add_1bank), E(fma_3banks), E(fma_2banks), E(fma_1bank), E(fma_same_reg), E(fma_abs_neg), E(fmac_2banks), E(fmac_1bank),
                E(bitop3_3banks), E(bitop3_1bank), E(bitop3_const), E(minu_2banks), E(mul_2banks), E(fmamk), E(sub_2banks),
                E(s1_v29), E(s1_v30), E(s1_v31), E(s1_v32), E(s1_v33), E(s1_v36), E(s2_v30), E(s2_v32), E(s2_v33), E(s2_v36), E(s2_v37), E(s12_same), E(bitop3_sgpr), E(bitop3_sgpr_same), E(add_sgpr), E(fmamk_2), E(fmac_sgpr), E(fma_3odd), E(fma_same_even), E(fma_sgpr_odd), E(fma_sgpr_mixed), E(fma_sgpr), E(fma_const)};
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  printf("%s, %d CUs; 8 waves/SIMD, %d x 64 instructions per wave; ns per instruction per SIMD (2.2 cycles = 0.94 ns)\n", prop.name, cus, iters);
  for (auto& e : es) {
    const int nb = cus * 8;
    hipLaunchKernelGGL(e.fn, dim3(nb), dim3(256), 0, 0, out, 1024);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL(e.fn, dim3(nb), dim3(256), 0, 0, out, iters);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms;
    CHECK(hipEventElapsedTime(&ms, a, b));
    printf("%-16s %8.3f ms  %6.3f ns/instr/SIMD\n", e.name, ms, ms * 1e6 / ((double)iters * 64 * 8));
  }
  return 0;
}
