// valu_clock.hip -- separates CLOCK from CYCLES in the VALU issue-cost measurement (tools/ubench/valu_cost.hip
// timed 0.3 ms kernels with events only, so "1.125 ns per v_add_f32 per SIMD" could be 2 cycles at 1.8 GHz or
// 2.7 cycles at 2.4 GHz).  Here every wave reads s_memtime (shader clock ticks, MI355X_MICROARCH.md) and the
// constant 100 MHz wall clock at its start and end, kernels run from 0.3 ms to > 50 ms (ITER is a run-time
// argument), at 8 and at 4 waves per SIMD.  Output per instruction: ns / instr / SIMD from events, cycles /
// instr / SIMD from s_memtime, and the clock the wave actually ran at = s_memtime ticks / wall-clock time.
// Build: hipcc --offload-arch=gfx950 -O2 -o valu_clock valu_clock.hip ; run: ./valu_clock
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

struct Stamp { unsigned long long cyc, wall; };

#define KERNEL32(NAME, ASM)                                                                              \
  __global__ void __launch_bounds__(256) k_##NAME(float* out, Stamp* st, float seed, int iters) {        \
    float r0 = seed, r1 = seed + 1, r2 = seed + 2, r3 = seed + 3, r4 = seed + 4, r5 = seed + 5, r6 = seed + 6, \
          r7 = seed + 7, x = seed * 0.5f + 1.0f, y = 1.0001f;                                             \
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), w0 = wall_clock64();                     \
    for (int i = 0; i < iters; i++) {                                                                    \
      asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)                               \
                   : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)      \
                   : "v"(x), "v"(y));                                                                    \
    }                                                                                                    \
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), w1 = wall_clock64();                     \
    out[blockIdx.x * 256 + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;                         \
    if ((threadIdx.x & 63) == 0) st[blockIdx.x * 4 + (threadIdx.x >> 6)] = Stamp{c1 - c0, w1 - w0};      \
  }
#define KERNEL64(NAME, ASM)                                                                              \
  __global__ void __launch_bounds__(256) k_##NAME(float* out, Stamp* st, float seed, int iters) {        \
    double r0 = seed, r1 = seed + 1, r2 = seed + 2, r3 = seed + 3, r4 = seed + 4, r5 = seed + 5, r6 = seed + 6, \
           r7 = seed + 7, x = seed * 0.5 + 1.0, y = 1.0001;                                               \
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), w0 = wall_clock64();                     \
    for (int i = 0; i < iters; i++) {                                                                    \
      asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)                               \
                   : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)      \
                   : "v"(x), "v"(y));                                                                    \
    }                                                                                                    \
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), w1 = wall_clock64();                     \
    out[blockIdx.x * 256 + threadIdx.x] = (float)(r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7);                 \
    if ((threadIdx.x & 63) == 0) st[blockIdx.x * 4 + (threadIdx.x >> 6)] = Stamp{c1 - c0, w1 - w0};      \
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

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  float* out;
  Stamp* st;
  const int max_blocks = cus * 8;
  CHECK(hipMalloc(&out, (size_t)max_blocks * 256 * sizeof(float)));
  CHECK(hipMalloc(&st, (size_t)max_blocks * 4 * sizeof(Stamp)));
  Stamp* hst = (Stamp*)malloc((size_t)max_blocks * 4 * sizeof(Stamp));
  Entry es[] = {E(add_f32, 1), E(mul_f32, 1), E(fma_f32, 1), E(min_f32, 1), E(mov_b32, 1), E(xor_b32, 1), E(rcp_f32, 1),
                E(cmp_cnd_pair, 2), E(add_f64, 1), E(fma_f64, 1), E(rcp_f64, 1)};
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  printf("%s, %d CUs, clockRate %d kHz\n", prop.name, cus, prop.clockRate);
  printf("%-14s %5s %9s %9s %13s %13s %9s\n", "instruction", "w/SIMD", "iters", "ms", "ns/instr/SIMD", "cyc/instr/SIMD", "clock GHz");
  const int iters_list[] = {4096, 65536, 786432};
  for (auto& e : es) {
    for (int wps = 8; wps >= 4; wps -= 4) {
      const int nb = cus * wps;  // wps blocks of 4 waves per CU = wps waves per SIMD, all resident at once
      for (int iters : iters_list) {
        if (wps == 4 && iters != 65536) continue;
        hipLaunchKernelGGL(e.fn, dim3(nb), dim3(256), 0, 0, out, st, 1.0f, iters > 65536 ? 65536 : iters);  // warm
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(a));
        hipLaunchKernelGGL(e.fn, dim3(nb), dim3(256), 0, 0, out, st, 1.0f, iters);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms;
        CHECK(hipEventElapsedTime(&ms, a, b));
        CHECK(hipMemcpy(hst, st, (size_t)nb * 4 * sizeof(Stamp), hipMemcpyDeviceToHost));
        double cyc = 0, wall = 0;
        for (int i = 0; i < nb * 4; i++) { cyc += (double)hst[i].cyc; wall += (double)hst[i].wall; }
        cyc /= nb * 4; wall /= nb * 4;  // mean per wave
        const double instr_wave = (double)iters * 8 * e.per_group;
        const double ns = ms * 1e6 / (instr_wave * wps);
        const double cpi = cyc / (instr_wave * wps);
        const double ghz = cyc / (wall * 10.0);  // wall clock ticks at 100 MHz = 10 ns
        printf("%-14s %5d %9d %9.3f %13.3f %13.3f %9.3f\n", e.name, wps, iters, ms, ns, cpi, ghz);
      }
    }
  }
  return 0;
}
