// pt_kernel.h -- host-visible interface of the device code in pt_kernel.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/ptcore.h"

#ifndef PT_BLOCK_THREADS
#define PT_BLOCK_THREADS 256
#endif
// one-lane-per-pixel kernels set their issue priority by progress from this many samples per pixel up (pt_kernel.hip)
#ifndef PT_PRIO_MIN_SPP
#define PT_PRIO_MIN_SPP 512
#endif
#ifndef PT_PRIO_MIN_SPP_REGEN  // the path-regeneration kernels (many-sphere scenes: a sample is long)
#define PT_PRIO_MIN_SPP_REGEN 32
#endif
#ifndef PT_FOOTPRINT_MIN_SPP
#define PT_FOOTPRINT_MIN_SPP 8  // the pixel-footprint analysis (pt_footprint.h) is done once per pixel: worth it from this many samples
#endif
#ifndef PT_SCREEN_UNROLL
#define PT_SCREEN_UNROLL 9  // requested unroll of the variant-2/4 screening loop (hipcc ignores it for runtime trip counts)
#endif
// Kernel variants (all bit-identical; DESIGN.md section 3 = which runs when, Appendix B.2 = how they came about): 0 literal, 1 lean FP64, 2 screened, 3 screened with packed
// FP32, 4 + cheap sqrt/rsqrt, 5 branch-free keys, 6 straight-line speculation (one lane per pixel), 7 two samples per
// lane, 8 four lanes per pixel, 9 two lanes per pixel, 10 per-lane path regeneration (open scenes), 11 = 10 + a
// conservative uniform grid over the small spheres (pt_grid.h), 12 = 11 with walk and shading decoupled per lane (lab), 13 = 11 with
// the sphere tests pooled across the lanes of a wave (pt_grid.h, "variant 13"), 14 = 13 with 1024-thread workgroups (large scenes).
#define PT_VARIANT_AUTO (-1)  // pt_renderer_opts_default(): resolved per launch by effective_variant() in pt_capi.hip
#define PT_DEFAULT_VARIANT 6  // the one-lane-per-pixel kernel the automatic policy uses when it does not pick variant 8
#ifndef PT_SCREEN_MAX_SPHERES
// Variants >= 5 use the key-based screen (every sphere fully evaluated, branch-free) up to this size and the
// many-sphere loop (float part, wave-uniform skip, scalar loads) above.  Measured crossover on random scenes
// with walls, 1024^2 x 16 spp (tools/threshold_sweep.py): 10 spheres 1.32 vs 1.36 ms, 12: 1.39 vs 1.32,
// 24: 2.02 vs 1.63, 64: 4.40 vs 2.81.  The key-based screen needs n <= 64 (index bits in the key).
#define PT_SCREEN_MAX_SPHERES 10
#endif
#ifndef PT_UNROLL_BOUNCES
#define PT_UNROLL_BOUNCES 1  // also emit a fully unrolled path for the reference's MAX_BOUNCES = 5 (+6 %)
#endif
#ifndef PT_UNROLL_NINE
#define PT_UNROLL_NINE 1  // also emit a fully unrolled screening pass for the reference's 9-sphere scene (+5 %)
#endif
#ifndef PT_KERNEL_ATTR
#define PT_KERNEL_ATTR  // e.g. __attribute__((amdgpu_waves_per_eu(4, 4))): let the scheduler spend registers on ILP
#endif
#ifndef PT_MIN_WAVES
#define PT_MIN_WAVES 4  // __launch_bounds__ 2nd argument: the register allocator must allow 4 waves per SIMD (<= 128 VGPRs)
#endif
#ifndef PT_REF_MIN_WAVES
#define PT_REF_MIN_WAVES 5  // ... of the reference-configuration builds of variant 6 (<= 96 VGPRs)
#endif
#ifndef PT_REF_MIN_WAVES_PHILOX
// ... the philox builds keep four.  Round 3 gave the 5-bounce philox build five (one spilled word, whole frame 48.7 -> 48.0 ms);
// round 4 measured what the scratch memory that build needs does to CHUNKED launches -- every workgroup of a kernel with
// scratch is slow to start, and a chunked tile is 6-8 times as many workgroups: a quarter frame (8 chunks) 13.8 ms against
// 12.2 with four waves and no scratch, a 64-row tile 13.6 against 6.6 -- for 0.7 % on the whole frame
// (profiles/r04/philox_waves.txt).
#define PT_REF_MIN_WAVES_PHILOX 4
#endif
// LDS the scene image may take per workgroup (gfx950 has 160 KiB per CU; one workgroup may
// use all of it; this budget still lets a full-size scene run with one workgroup per CU and the
// 9-sphere scene with many).
#define PT_LDS_BUDGET_BYTES (104 * 1024)
// Variant 11 (uniform grid, pt_grid.h) stages the geometry of every sphere (16 B each) beside its tables; it pays from
// about 160 spheres (16 spp, variant 10 vs 11: 120 spheres + walls 4.56 vs 5.22 ms, 180: 6.15 vs 5.99, 1000: 32.1 vs 9.3;
// without walls 120: 1.24 vs 1.04, 180: 1.92 vs 1.22; tools/many_ab.py).
#ifndef PT_GRID_BLOCK_THREADS
// The grid kernel (variant 11) is latency-bound -- dependent LDS reads and a long dependency chain per sphere test -- so it
// wants WAVES, and its 36 KB LDS image (geometry + tables at 1000 spheres) is per workgroup: 256-thread workgroups stop at
// four per CU (4 waves per SIMD).  512-thread workgroups share one image between eight waves; with the register cap below
// six waves per SIMD fit (three workgroups, 108 KB).  Measured at 1000 spheres, 32 spp: 256/4 27.0 ms, see HISTORY.md B.3.
#define PT_GRID_BLOCK_THREADS 512
#endif
// "variant 14" (round 5) = variant 13 with 1024-thread workgroups: one per CU (the same four waves per SIMD), ONE grid image instead of
// two, so the cell table of a scene above ~1200 spheres gets the other half of the CU's LDS (pt_grid.h, grid_max_entries)
#define PT_GRID_WIDE_THREADS 1024
#define PT_LDS_WIDE_BUDGET_BYTES (159 * 1024)
#ifndef PT_GRID_WIDE_MIN_SPHERES
#define PT_GRID_WIDE_MIN_SPHERES 1200
#endif
#ifndef PT_GRID_MIN_WAVES
#define PT_GRID_MIN_WAVES 6  // __launch_bounds__ 2nd argument of the grid kernel: <= 80 VGPRs
#endif
// Variant 12 (variant 11 with the walk and the shading decoupled per lane, pt_kernel.hip): a wave leaves the walk when this
// many of its lanes have nothing left to do there
#ifndef PT_GRID_PARK
#define PT_GRID_PARK 48
#endif
#ifndef PT_GRID12_MIN_WAVES
#define PT_GRID12_MIN_WAVES 4
#endif
#ifndef PT_POOL_HELPERS
#define PT_POOL_HELPERS 1  // variant 13: lanes whose pixel is finished stay in the loop as helpers of the wave's pooled sphere tests (pt_kernel.hip)
#endif
#ifndef PT_V13_WELFORD_TABLE
#define PT_V13_WELFORD_TABLE 1  // variant 13: the Welford updates share one count and take delta / n through the 1/n table like the reference-scene kernels
#endif
#ifndef PT_V13_DEAD_END
#define PT_V13_DEAD_END 1  // variant 13: the last bounce of a path forms no next ray (pt_trace.h) and walks only if it may end on an emitting sphere (pt_grid.h)
#endif
#ifndef PT_POOL_MIN_WAVES
// variant 13 (variant 11 with the sphere tests pooled across the wave, pt_grid.h) is issue-bound, not latency-bound like its
// predecessor: four waves per SIMD with 128 registers and no spills beat six with 80 and 128 B of scratch (13.9 -> 13.3 ms)
#define PT_POOL_MIN_WAVES 4
#endif
#define PT_GRID_MAX_SPHERES 2048
#define PT_GRID_MIN_SPHERES 160
// ... and on tiles that fill the chip (three waves per SIMD and more) already from 72 spheres: with the pooled tests, the sweep and
// the helpers the grid kernel passes the brute-force one between 60 and 80 spheres (1024^2 x 32 spp, closed / open, variant 10
// against 13: 60 spheres 5.01 / 1.08 against 5.16 / 1.17 ms, 80: 6.09 / 1.57 against 5.46 / 1.40, 140: 9.43 / 2.81 against
// 6.18 / 1.79; tools/grid_threshold.py).  Small tiles keep the four-lane / regeneration kernels up to 159.
#ifndef PT_GRID_MIN_SPHERES_LARGE_TILE
#define PT_GRID_MIN_SPHERES_LARGE_TILE 72
#endif

// Everything pixel_kernel needs travels as kernel arguments (SGPRs): the camera is 60 B, so
// the reference's two per-frame cudaMemcpy H2D (Renderer.h:59-60) disappear.
struct PixelKernelArgs {
  float* out;                  // first float of row `row_begin`, layout [row][col][14]
  const pt_sphere* spheres;    // device, reference 40-byte AoS layout
  uint32_t* rng_state;         // xorwow: 6 words per tile pixel, or nullptr (fresh generator)
  float basis[12];             // Camera::getEyeRayBasis corners
  float eye[3];
  int32_t n_spheres;
  int32_t width, height;       // image
  int32_t row_begin;           // tile origin
  uint32_t tile_pixels;        // (row_end - row_begin) * width
  int32_t spp;
  int32_t max_bounces;
  uint32_t frame;
  uint32_t scene_lds_f4;       // float4 slots of the LDS scene image (filled in by the launcher)
  uint32_t planar;             // 1 = channel-first output [14][tile rows][width] instead of [row][col][14]
  uint32_t* fail_count;        // variant 8: number of pixels whose speculation failed (may be nullptr)
  const uint32_t* accel;       // variant 11: the grid built by build_grid_kernel for this frame's scene
  uint64_t seed;
  // sample chunking (reference-configuration variant 6 on frames that make few rounds of workgroups, pt_kernel.hip):
  // a pixel block's samples are split over `chunks` workgroups of one launch, chained through chunk_state / chunk_flag
  uint32_t chunks;             // 0 or 1 = off
  uint32_t* chunk_state;       // PT_CHUNK_WORDS words per tile pixel, [word][pixel]
  uint32_t* chunk_flag;        // per pixel block: number of chunks completed (zeroed before the launch)
  uint32_t* err_word;          // device error word of the renderer (PT_DEVERR_*), OR-ed into by a kernel that cannot go on correctly
  uint64_t chunk_wait_ticks;   // wall-clock ticks (hipDeviceAttributeWallClockRate) a chunk waits for its predecessor before it gives up
  uint32_t debug;              // lab library only (PT_DEBUG_*): deliberate faults for the failure-path tests
  uint32_t prio;               // 1: one-lane-per-pixel waves set their issue priority by progress (filled in by the launcher)
  float* vertices;             // non-null: the frame's display vertices as well -- denoise_kernel's (col, width - row, RGBA8 packed in a
                               // float) per pixel, [tile row][col][3], written from the registers that hold the colour (pt_renderer_set_display)
  uint32_t repair;             // != 0: repair launch after a broken chunk chain -- unchunked, and pixel blocks whose chunk_flag equals
                               // this value (= the chunk count of the broken launch: complete) are skipped
};
// A batch of frames in one launch (pt_renderer_enqueue_frames): the single-frame arguments plus, per frame, the camera
// (basis[12], eye[3]) and where the frame goes.  At most PT_FRAMES_MAX frames per launch (the argument segment is 4 KB).
#define PT_FRAMES_MAX 32
struct FramesKernelArgs {
  PixelKernelArgs base;        // out / vertices: frame 0's; frame: frame 0's number (philox key)
  uint32_t frames;             // 2 .. PT_FRAMES_MAX
  uint64_t out_stride;         // floats from one frame's buffer to the next one's
  uint64_t vtx_stride;         // ... of the display vertices (fused display pack)
  float cams[PT_FRAMES_MAX][15];
};
// words handed from one chunk of a pixel block to the next: 10 sums, 2 counts (colour; the three first-hit accumulators share
// one), 4 x {mean, M2}, and the 6 generator words (xorwow only; philox needs none)
#define PT_CHUNK_WORDS 26
#ifndef PT_CHUNKS
// chunks per pixel block of the automatic policy.  Measured at the headline frame (tools/chunk_sweep.py, profiles/r03):
// 1 (off) 50.88 ms, 2: 50.14, 4: 50.22, 8: 50.07, 16: 50.23 -- two chunks take nearly all of the gain for one hand-over per
// pixel (218 MB of HBM traffic per frame instead of the 1.5 GB seven hand-overs cost)
#define PT_CHUNKS 2
#endif
#define PT_CHUNKS_TILE 8        // ... on tiles of at most 8.5 one-lane waves per SIMD (half a 1024^2 frame and less), pt_capi.hip
#define PT_CHUNKS_SMALL_TILE 6  // ... and of at most 3.25
#define PT_CHUNKS_SPLIT 4       // ... in the split kernels (variants 8, 9)
#define PT_CHUNKS_MAX 16
#ifndef PT_CHUNKS_GRID_MAX
// ... of the pooled grid kernel: round 5's kernel is flat from 4 to 8 chunks (closed / open at 256 spp, 4 | 6 | 8: 71.4 | 71.2 | 71.3 and
// 22.4 | 22.3 | 22.6 ms, profiles/r05/cfg4_ab.txt) and every hand-over is 208 B of traffic per pixel: four
#define PT_CHUNKS_GRID_MAX 4
#endif
// A chunk may only be as long as keeps the worst chained wait (all chunks of a block co-resident: chunk k waits k chunk
// durations) far inside the wait limit: at most this many samples per chunk (about 50 ms of kernel time on an MI355X)
#define PT_CHUNK_MAX_SAMPLES 4096
// ... of the pooled grid kernel (variant 13), whose samples are a hundred times as long (about 0.1 ms per sample and 512-pixel
// workgroup at 1000 spheres, three times that at 2048 spheres and 8 bounces): 16 co-resident chunks of 256 samples stay
// near a second, a quarter of the default wait limit
#define PT_CHUNK_MAX_SAMPLES_GRID 256
#define PT_DEVERR_CHUNK_CHAIN 1u  // a chunk's predecessor never signalled: the frame is incomplete
#define PT_CHUNK_FAILED 0x80000000u  // chunk_flag bit: the hand-over chain of this pixel block is broken (pt_kernel.hip, chunk_wait)
#define PT_DEBUG_DROP_CHUNK_FLAG 1u  // pixel block 0, chunk 0 does not publish its flag (tests/test_chunk_chain_gpu.py)

#ifndef PT_BUILD_EXPERIMENTS
#define PT_BUILD_EXPERIMENTS 0  // 1: also build variants 1-5, 7, 9, 12 (libptcore_lab.so)
#endif
#define PT_VARIANT_FAST 100     // reported by pt_renderer_kernel_info for a fast_math renderer (pt_fast.hip)
#define PT_FAST_LDS_SPHERES 64  // the fast kernel stages scenes up to this size into LDS, larger ones are read in place
int pt_kernel_num_variants(void);
bool pt_kernel_chunked(int variant, int n_spheres, int max_bounces, bool planar, int spp, uint32_t chunks);
int pt_kernel_ref_bounces(int n_spheres, int max_bounces, int variant, bool planar);  // bounce cap of the reference-configuration build a launch runs, 0 = generic build
int pt_kernel_block_threads(int variant);  // workgroup size the launcher uses
bool pt_kernel_has_variant(int variant);  // compiled into this library?
const void* pt_kernel_symbol(int rng_mode, int variant, int n_spheres, int max_bounces, bool planar);  // the function a launch with these parameters runs
size_t pt_kernel_lds_bytes(int n_spheres, int variant);
int pt_kernel_max_spheres(int variant);
hipError_t pt_launch_pixel_kernel(const PixelKernelArgs& a, int rng_mode, int variant, hipStream_t stream);
// frame batches: is there a kernel for these launch parameters (reference scene, variant 6, interleaved layout), and the launch
bool pt_kernel_has_frames(int variant, int n_spheres, int max_bounces, bool planar);
hipError_t pt_launch_frames_kernel(const FramesKernelArgs& fa, int rng_mode, hipStream_t stream);
hipError_t pt_launch_build_grid(const pt_sphere* spheres, int n, uint32_t* accel, const float* eye /* camera hint or NULL */,
                                bool pooled /* for variant 13's LDS image (fewer cells at large n) */, hipStream_t stream,
                                int threads = PT_GRID_BLOCK_THREADS /* workgroup size of the kernel that will stage the grid */);
void pt_kernel_grid_layout(int n_spheres, int threads, uint64_t out[8]);  // image bytes, part offsets, table capacities (pt_kernel.hip)
size_t pt_kernel_accel_bytes(void);  // device scratch a renderer must provide in PixelKernelArgs::accel for variant 11
hipError_t pt_launch_setup_random(uint32_t* state, int width, int row_begin, uint32_t tile_pixels, uint64_t seed,
                                  hipStream_t stream);
// toleranced fast mode (pt_fast.hip)
const void* pt_fast_kernel_symbol(int rng_mode, int n_spheres, int max_bounces);
size_t pt_fast_kernel_lds_bytes(int n_spheres);
hipError_t pt_launch_fast_kernel(const PixelKernelArgs& a, int rng_mode, hipStream_t stream);
