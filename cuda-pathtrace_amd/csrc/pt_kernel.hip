// pt_kernel.hip -- the per-pixel Monte-Carlo megakernel for gfx950 (MI355X).
//
// Replaces src/pathtrace.cu's pixel_kernel (:203-257) and setup_random (:259-266).
// One lane = one pixel, lanes of a wave64 are 64 CONSECUTIVE COLUMNS of one image row
// (the reference maps adjacent threads to adjacent rows, i.e. a 56*W-byte lane stride).
// The sphere list is staged once per workgroup into LDS; generator state, ray, throughput,
// AOV sums and the four Welford accumulators live in VGPRs for the whole frame.
#include "pt_trace.h"

#pragma clang fp contract(off)

namespace pt {

// pixel_kernel: src/pathtrace.cu:203-257
// LEAN selects the geometry-only LDS layout of many-sphere scenes (pt_scene_lds.h) at compile time, so the
// 9-sphere kernels carry no trace of it.
// REFB != 0 builds the kernel for the reference's own scene size -- 9 spheres (Scene.h:23) -- and a fixed bounce cap as
// compile-time constants: REFB = 5 is the reference's MAX_BOUNCES (pathtrace.cu:7), REFB = 8 the interactive configuration
// (BASELINE.json configs[4]).  No generic loops, no index-width arithmetic, and a hot loop that is a third smaller.
template <int VAR, bool WIDE = false>
constexpr int kBlockThreads = (VAR == 13 && WIDE) ? PT_GRID_WIDE_THREADS : (VAR == 11 || VAR == 12 || VAR == 13) ? PT_GRID_BLOCK_THREADS : PT_BLOCK_THREADS;
template <int VAR>
constexpr int kMinWaves = (VAR == 11) ? PT_GRID_MIN_WAVES : (VAR == 13) ? PT_POOL_MIN_WAVES : (VAR == 12) ? PT_GRID12_MIN_WAVES : PT_MIN_WAVES;

// the reference-configuration builds of variant 6 with the XORWOW generator fit 96 registers (12 bytes of spills, in cold code:
// WRITE_SIZE stays at the algorithmic bytes): five waves per SIMD instead of four (headline frame 50.12 -> 49.81 ms, three
// alternating runs each, profiles/r03/README.md).  The philox builds spilled warm then (WRITE_SIZE 143 -> 400 MB per frame) and
// gained nothing; after the bit-operation work of round 3 the 5-bounce build fits 96 registers with ONE spilled word (a reload
// per sample) and gains 1.5 % (48.73 -> 47.98 ms, profiles/r03/README.md); the 8-bounce build (4 words) does not (0.0963 -> 0.0971 ms)
// and keeps four waves, like every other build.
template <int VAR, int REFB, int RNG>
constexpr int kMinWavesR = (VAR == 6 && REFB != 0) ? (RNG == PT_RNG_XORWOW ? PT_REF_MIN_WAVES : (REFB == 5 ? PT_REF_MIN_WAVES_PHILOX : kMinWaves<VAR>)) : kMinWaves<VAR>;

// builds that can chain a pixel's samples through several workgroups of one launch (sample chunking, below): the reference-
// configuration builds of variant 6 and the pooled grid kernel
template <int VAR, int REFB>
constexpr bool kChunkable = (VAR == 6 && REFB != 0) || VAR == 13;

// ---- the hand-over chain of sample chunking (what it is for: pixel_kernel below) ------------------------------------------
// chunk_flag[block]: bits 0-30 = chunks of the pixel block that have completed, bit 31 (PT_CHUNK_FAILED) = the chain is broken.
// Wait for the predecessor of chunk `chunk`.  The wait is bounded in TIME (s_memrealtime) and giving up is an ERROR the host
// sees (device error word -> PT_EKERNEL or a repair launch, pt_capi.hip), never a silent frame.  A workgroup that gives up, or
// finds the chain already broken, marks the chain broken and leaves WITHOUT touching the frame, the hand-over buffer or the
// generator state -- so do all its successors (they can only ever see the mark: the count stops below them), the last chunk
// included, and the last chunk is the only one that writes the frame and the generator state.  Hence after a launch every
// pixel block is either complete (count == chunks: frame and state exactly the reference's) or untouched (frame and state as
// before the launch), which is what lets the host render the untouched blocks again, unchunked (a.repair).
// Returns true (workgroup-uniform) when the caller has to leave.
__device__ __forceinline__ bool chunk_wait(const PixelKernelArgs& a, uint32_t block_id, uint32_t chunk) {
  __shared__ uint32_t s_pred;
  if (threadIdx.x == 0) {
    const uint64_t t0 = wall_clock64();
    uint32_t spins = 0, f;
    for (;;) {
      f = __hip_atomic_load(a.chunk_flag + block_id, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
      if ((f & PT_CHUNK_FAILED) != 0u || f >= chunk) break;
      __builtin_amdgcn_s_sleep(32);
      if ((++spins & 255u) == 0u && wall_clock64() - t0 > a.chunk_wait_ticks) {
        f = PT_CHUNK_FAILED;
        break;
      }
    }
    if ((f & PT_CHUNK_FAILED) != 0u) {
      if (a.err_word) atomicOr(a.err_word, PT_DEVERR_CHUNK_CHAIN);
      __hip_atomic_fetch_or(a.chunk_flag + block_id, PT_CHUNK_FAILED, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    s_pred = f;
  }
  __syncthreads();
  return (s_pred & PT_CHUNK_FAILED) != 0u;
}

// chunk `chunk` of the block is complete: everything it stored is visible device-wide before the count says so.  The count
// only grows and never clears the failure mark (a predecessor that was merely slow may finish after its successor gave up).
__device__ __forceinline__ void chunk_publish(const PixelKernelArgs& a, uint32_t block_id, uint32_t chunk) {
  __syncthreads();  // every wave's stores are issued ...
  if (threadIdx.x == 0) {
    __threadfence();  // ... and visible device-wide before the flag says so
#if PT_BUILD_EXPERIMENTS  // lab library: a deliberately broken chain for the failure-path tests
    if ((a.debug & PT_DEBUG_DROP_CHUNK_FLAG) && block_id == 0u && chunk == 0u) return;
#endif
    __hip_atomic_fetch_max(a.chunk_flag + block_id, chunk + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// FRAMES (round 5, pt_launch_frames_kernel below): the launch renders a BATCH of frames with known cameras (which travel as
// kernel arguments too), each into its own buffer, starting from zeroed sums.  Per pixel the operations and their order are those
// of fr_count single-frame launches.  What links two frames is the generator (the reference's d_states carries every pixel's
// XORWOW state from one Render() to the next: pathtrace.cu:212,256), hence two shapes:
//  * XORWOW: a workgroup keeps its pixel block for the whole batch and LOOPS over the frames; the generator simply stays in
//    its registers.  What goes is the ramp and tail of fr_count launches, the per-launch staging and the state traffic.  (First
//    built as one workgroup per (frame, block) with the state handed on through the sample chunks' flags: a workgroup that
//    becomes resident while its predecessor still runs only waits, and the dispatcher cannot pass it over for one that is
//    ready -- 0.1145 ms per frame against 0.097 as single launches; profiles/r05/cfg5_frames.txt.  Also measured, no effect:
//    scattering the pixel blocks over the workgroups, so that no CU holds four neighbouring blocks -- 0.0936 vs 0.0938.)
//  * the counter-based generator is re-keyed per frame, so its frames are independent: one workgroup per (frame, pixel block),
//    frame-major -- the next frame's workgroups fill the slots the current one's leave (and the fifth slot per CU a single
//    1024-workgroup frame cannot use).
template <bool FRAMES> struct KernelArgsOf { typedef PixelKernelArgs type; };
template <> struct KernelArgsOf<true> { typedef FramesKernelArgs type; };  // (the cameras travel as kernel arguments too)
template <bool FRAMES> __device__ __forceinline__ PixelKernelArgs& base_args(typename KernelArgsOf<FRAMES>::type& args) {
  if constexpr (FRAMES) return args.base; else return args;
}

// WIDE (variant 13 only; the launcher's "variant 14"): 1024-thread workgroups, one per CU -- pt_grid.h, grid_max_entries
template <int RNG, int VAR, bool LEAN = false, int REFB = 0, bool FRAMES = false, bool WIDE = false>
__global__ void __launch_bounds__((kBlockThreads<VAR, WIDE>), ((FRAMES && RNG == PT_RNG_XORWOW) ? PT_MIN_WAVES : kMinWavesR<VAR, REFB, RNG>)) PT_KERNEL_ATTR
pixel_kernel(typename KernelArgsOf<FRAMES>::type args) {  // (an XORWOW batch has one workgroup per pixel block for all its frames: the
                                                           // interactive shape fills four of a CU's five slots, so that build takes 128 registers)
  PixelKernelArgs& a = base_args<FRAMES>(args);
  // a batch of frames, two shapes (head of this section): the counter-based generator's frames are independent -- one workgroup
  // per (frame, pixel block), frame-major; XORWOW's are a chain per pixel -- one workgroup per pixel block, looping over the frames
  constexpr bool FRAME_GRID = FRAMES && RNG == PT_RNG_PHILOX, FRAME_LOOP = FRAMES && !FRAME_GRID;
  uint32_t fr = 0u, fr_count = 1u, first_block = blockIdx.x;
  auto load_camera = [&](uint32_t f) {
    if constexpr (FRAMES) {
      // by scalar loads from the argument segment itself (indexing the by-value struct with a run-time index would make the
      // compiler copy all of it to scratch memory)
      typedef const __attribute__((address_space(4))) float* cfloats;
      typedef const __attribute__((address_space(4))) char* cbytes;
      const cfloats cam = (cfloats)((cbytes)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(FramesKernelArgs, cams)) + 15u * f;
#pragma unroll
      for (int k = 0; k < 12; k++) a.basis[k] = cam[k];
#pragma unroll
      for (int k = 0; k < 3; k++) a.eye[k] = cam[12 + k];
    }
  };
  if constexpr (FRAMES) fr_count = args.frames;
  if constexpr (FRAME_GRID) {
    const uint32_t n_blocks = (a.tile_pixels + kBlockThreads<VAR, WIDE> - 1) / kBlockThreads<VAR, WIDE>;
    fr = blockIdx.x / n_blocks;  // (workgroup-uniform)
    first_block = blockIdx.x - fr * n_blocks;
    load_camera(fr);  // before the scene image is staged for this frame's eye
  }
  constexpr bool REF = REFB != 0;
  constexpr bool CHUNKS = kChunkable<VAR, REFB> && !FRAMES;
  static_assert(!FRAMES || (VAR == 6 && REFB != 0), "frame batches: reference-configuration builds of variant 6");
  if constexpr (REF) {
    a.n_spheres = 9;
    a.max_bounces = REFB;
  }
  if constexpr (CHUNKS) {
    // repair launch (pt_capi.hip, after a chunked launch whose chain broke): unchunked, and only for the pixel blocks that
    // launch left untouched -- the ones it completed are not rendered a second time
    if (a.repair != 0u && __hip_atomic_load(a.chunk_flag + blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a.repair) return;
  }
  extern __shared__ float4 lds_scene[];
  SceneLds sc = stage_scene<VAR == 3>(a.spheres, a.n_spheres, lds_scene, LEAN, mk3(a.eye[0], a.eye[1], a.eye[2]), a.spp);
  // variants with a lean build run their LDS build only on scenes up to PT_SCREEN_MAX_SPHERES (launcher): the
  // many-sphere path is not even compiled into it, which keeps the hot loop's code small
  sc.small_only = !LEAN && (VAR == 6 || VAR == 8 || VAR == 10);
  constexpr bool kRegen = (VAR == 10 || VAR == 11 || VAR == 13);
  GridLds grid;
  if constexpr (VAR == 11 || VAR == 12 || VAR == 13) {  // the frame's grid, built by build_grid_kernel just before this launch
    grid = stage_grid<VAR == 13>(a.spheres, a.n_spheres, a.accel, lds_scene + a.scene_lds_f4);  // after the two small tables
    sc.grid = &grid;
    if constexpr (VAR == 13)  // the test pool of each wave follows the grid image (16-byte aligned)
      sc.pool = reinterpret_cast<char*>(lds_scene + a.scene_lds_f4) + ((grid_lds_bytes(a.n_spheres, true, kBlockThreads<VAR, WIDE>) + 15) & ~(size_t)15);
  }

  // Sample chunking (REF builds of variant 6, variant 13): workgroup blockIdx.x = chunk * n_blocks + block renders samples
  // [chunk * per, (chunk + 1) * per) of its 256 pixels and hands generator, sums and Welford accumulators to the next chunk
  // through chunk_state.  Forward progress (DESIGN.md section 3 ("sample chunking")): every dispatcher hands out its workgroups in
  // increasing index order, so by induction on the lowest unfinished index the predecessor of a waiting workgroup is resident
  // or done -- and the lowest unfinished one never waits.  That is a property of today's hardware, not of the programming
  // model, so the wait is bounded in time and giving up raises the renderer's error word (PT_EKERNEL), never a silent frame.
  // Same operations per pixel in the same order; what changes is that a frame of few, long workgroups becomes one of many
  // short ones (tools/shape_sweep.py: why).
  uint32_t block_id = FRAMES ? first_block : blockIdx.x, chunk = 0u, n_chunks = 1u;
  if constexpr (CHUNKS) {
    if (a.chunks > 1u) {
      const uint32_t n_blocks = (a.tile_pixels + kBlockThreads<VAR, WIDE> - 1) / kBlockThreads<VAR, WIDE>;
      n_chunks = a.chunks;
      chunk = blockIdx.x / n_blocks;
      block_id = blockIdx.x - chunk * n_blocks;
    }
  }
  const uint32_t tp = block_id * kBlockThreads<VAR, WIDE> + threadIdx.x;  // pixel index inside the tile
  const bool active = tp < a.tile_pixels;  // lanes past the tile stay for the cooperative epilogue
  const int row = a.row_begin + (int)(tp / (uint32_t)a.width);
  const int col = (int)(tp % (uint32_t)a.width);
  const uint32_t id = (uint32_t)row * (uint32_t)a.width + (uint32_t)col;  // :206

  Rng<RNG> rng;
  if constexpr (RNG == PT_RNG_XORWOW) {
    if (a.rng_state && active) {  // :212
      const uint32_t* s = a.rng_state + (size_t)tp * 6;
      rng.st = Xorwow{s[0], s[1], s[2], s[3], s[4], s[5]};
    } else {
      xorwow_init(rng.st, (uint64_t)id + a.seed);  // :265
    }
  } else {
    rng.k0 = (uint32_t)a.seed;
    rng.k1 = (uint32_t)(a.seed >> 32) ^ a.frame;
    rng.pix = id;
  }

  const bool pow2_image = ((a.width & (a.width - 1)) == 0) && ((a.height & (a.height - 1)) == 0);  // wave-uniform
  const float inv_w = 1.0f / (float)a.width, inv_h = 1.0f / (float)a.height;  // exact for powers of two

  // (the frame loop of a batch is a backward goto that exists in the FRAME_LOOP builds only: every other build's code is,
  // instruction for instruction, what it was without it -- a `for` with one trip was not, and the headline kernel's allocation is touchy)
frame_top:
  if constexpr (FRAME_LOOP) {
    load_camera(fr);
    if (fr > 0u) {  // the scene image's eye part (SceneLds::eyeg: off and c of :73,76 for rays from THIS frame's eye), as stage_scene forms it
      __syncthreads();  // every wave has finished the previous frame's primary rays
      for (int i = threadIdx.x; i < a.n_spheres; i += blockDim.x) {
        const float4 g = sc.geom[i];
        const F3 off = mk3(a.eye[0] - g.x, a.eye[1] - g.y, a.eye[2] - g.z);
        sc.eyeg[i] = make_float4(off.x, off.y, off.z, dot(off, off) - g.w);
      }
      __syncthreads();
    }
  }
  if constexpr (FRAMES && RNG == PT_RNG_PHILOX) rng.k1 = (uint32_t)(a.seed >> 32) ^ (a.frame + fr);
  const F3 B0 = mk3(a.basis[0], a.basis[1], a.basis[2]), B1 = mk3(a.basis[3], a.basis[4], a.basis[5]);
  const F3 B2 = mk3(a.basis[6], a.basis[7], a.basis[8]), B3 = mk3(a.basis[9], a.basis[10], a.basis[11]);
  const F3 eye = mk3(a.eye[0], a.eye[1], a.eye[2]);

  Welford var[4] = {{0, 0.0f, 0.0f}, {0, 0.0f, 0.0f}, {0, 0.0f, 0.0f}, {0, 0.0f, 0.0f}};
  TraceOutput L{mk3(0, 0, 0), mk3(0, 0, 0), mk3(0, 0, 0), 0.0f};
  int i_begin = 0, i_end = a.spp;
  if constexpr (CHUNKS) {
    if (n_chunks > 1u) {
      const int per = (a.spp + (int)n_chunks - 1) / (int)n_chunks;
      i_begin = (int)chunk * per;
      i_end = i_begin + per < a.spp ? i_begin + per : a.spp;
      if (chunk > 0u) {
        if (chunk_wait(a, block_id, chunk)) return;  // the chain of this pixel block is broken: nothing of the block is written
        if (active) {
          auto ld = [&](int w) { return __hip_atomic_load(a.chunk_state + (size_t)w * a.tile_pixels + tp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
          auto ldf = [&](int w) { return __uint_as_float(ld(w)); };
          L.color = mk3(ldf(0), ldf(1), ldf(2));
          L.normal = mk3(ldf(3), ldf(4), ldf(5));
          L.albedo = mk3(ldf(6), ldf(7), ldf(8));
          L.depth = ldf(9);
          const int n0 = (int)ld(10), n1 = (int)ld(11);  // the three first-hit accumulators count together (welford_update3)
#pragma unroll
          for (int k = 0; k < 4; k++) var[k] = Welford{k == 0 ? n0 : n1, ldf(12 + 2 * k), ldf(13 + 2 * k)};
          if constexpr (RNG == PT_RNG_XORWOW) rng.st = Xorwow{ld(20), ld(21), ld(22), ld(23), ld(24), ld(25)};
        }
      }
    }
  }

  auto primary_ray = [&](Rng<RNG>& g, F3& dir) {  // :221-229
    float sx = (float)row, sy = (float)col;
    if (a.spp != 1) {
      float jx, jy;
      g.jitter(jx, jy);
      sx += jx * 1.0f - 0.5f;
      sy += jy * 1.0f - 0.5f;
    }
    if (pow2_image) {  // x / 2^k == x * 2^-k exactly: spares two correctly rounded divisions per sample
      sx *= inv_h;
      sy *= inv_w;
    } else {
      sx /= (float)a.height;  // contract C7
      sy /= (float)a.width;
    }
    dir = lerp(lerp(B0, B1, sy), lerp(B2, B3, sy), 1.0f - sx);
  };

  // the primary direction for screen position (sx, sy) in pixel units, by primary_ray's own formula (footprint analyses)
  auto dir_at = [&](float sx, float sy) {
    if (pow2_image) {
      sx *= inv_h;
      sy *= inv_w;
    } else {
      sx /= (float)a.height;
      sy /= (float)a.width;
    }
    return lerp(lerp(B0, B1, sy), lerp(B2, B3, sy), 1.0f - sx);
  };
  if constexpr (REF && VAR == 6) {
#ifndef PT_NO_FOOTPRINT
    // once per pixel: the spheres this pixel's primary rays can return; the wave ranks the union at bounce 0
    if (a.spp >= PT_FOOTPRINT_MIN_SPP) {
      const uint32_t mine = active ? primary_candidates(sc, a.n_spheres, (float)row, (float)col, dir_at) : 0u;
      uint32_t wave_mask = 0u;
#pragma unroll
      for (int j = 0; j < 9; j++) wave_mask |= (__builtin_amdgcn_ballot_w64(((mine >> j) & 1u) != 0u) != 0ull) ? (1u << j) : 0u;
      sc.prim_mask = wave_mask;
    }
#endif
  }

  int i = active ? i_begin : i_end;  // inactive lanes trace nothing ([i_begin, i_end): all samples unless the frame is chunked)
  bool prim_ok = false;  // variant 13: this pixel's primary rays take their spheres from the pixel's list instead of walking the grid
  if constexpr (VAR == 13 && PT_PRIMLIST) {
    // once per pixel and workgroup: the grid spheres the pixel's primary rays can return (pt_primlist.h)
    if (grid.valid && a.spp >= PT_PRIMLIST_MIN_SPP && a.max_bounces >= 1) {
      int k = 0;
      const uint32_t slot = grid.h.prim_base + (uint32_t)kPrimEntriesPerLane * threadIdx.x;
      prim_ok = build_primary_list(grid, a.spheres, a.n_spheres, eye, (float)row, (float)col, active, dir_at, pool_of_wave(sc.pool).ring,
                                   const_cast<uint2*>(grid.cells), slot, k);
      // a pixel that sees nothing at all: every sample is `output.color += color; return` (:157-161) after its jitter draws
      if (prim_ok & (k == 0) & ((grid.h.n_big & 0xFFFFu) == 0u)) {
        for (; i < i_end; i++) {
          if constexpr (RNG == PT_RNG_XORWOW) {  // (philox: begin_sample only positions the counter)
            if (a.spp != 1) {
              float jx, jy;
              rng.jitter(jx, jy);  // :223-224
            }
          }
          L.color = L.color + mk3(0.0f, 0.0f, 0.0f);  // :159
        }
      }
    }
  }
  if constexpr (kRegen) {
    // Path regeneration (the bit-exact form of active-ray compaction for a kernel whose accumulators are
    // per lane): the sample loop and the bounce loop are flattened into one per-lane state machine, so a
    // lane whose path left the scene starts its next sample at once instead of idling until the longest
    // path of the wave has finished its bounces.  Every lane still executes exactly the reference's
    // sequence for its pixel (same draws, same sums, same Welford updates, same order); only the
    // alignment BETWEEN lanes changes.  Pays in open scenes, costs the unrolled path in closed ones.
    int n = 0;  // depth of the current path; 0 = start a new sample
    F3 o = eye, d = eye;
    F3 color = mk3(0.0f, 0.0f, 0.0f), mask = mk3(1.0f, 1.0f, 1.0f);
    // issue priority by progress, as in the one-lane loop below (there: why); the progress of a wave is that of its first lane
    const int span = i_end - i_begin;  // this workgroup's samples
    const bool by_progress = span >= PT_PRIO_MIN_SPP_REGEN;
    const int q1 = i_begin + span / 4, q2 = i_begin + span / 2, q3 = i_end - span / 4;
    int last_band = -1;
    // (variant 13: the wave stays together until its LAST lane is done -- lanes that have finished their samples keep going through
    // the nearest-hit search as helpers of the pooled sphere tests, which otherwise run ever emptier towards the end of every workgroup)
    for (;;) {
      const bool live = i < i_end;
      if constexpr (VAR == 13 && PT_POOL_HELPERS) {
        if (__builtin_amdgcn_ballot_w64(live) == 0) break;
      } else {
        if (!live) break;
      }
      if (by_progress) {
        int iu;  // the first lane that still works
        if constexpr (VAR == 13 && PT_POOL_HELPERS) iu = __builtin_amdgcn_readlane(i, __builtin_ctzll(__builtin_amdgcn_ballot_w64(live)));
        else iu = __builtin_amdgcn_readfirstlane(i);
        const int band = (iu >= q1 ? 1 : 0) + (iu >= q2 ? 1 : 0) + (iu >= q3 ? 1 : 0);
        if (band != last_band) {  // wave-uniform
          last_band = band;
          if (band == 0) __builtin_amdgcn_s_setprio(3);
          else if (band == 1) __builtin_amdgcn_s_setprio(2);
          else if (band == 2) __builtin_amdgcn_s_setprio(1);
          else __builtin_amdgcn_s_setprio(0);
        }
      }
      if (live & (n == 0)) {  // :219-229
        rng.begin_sample((uint32_t)i);
        primary_ray(rng, d);
        o = eye;
        color = mk3(0.0f, 0.0f, 0.0f);
        mask = mk3(1.0f, 1.0f, 1.0f);
      }
      bool escaped = false;
      if (n < a.max_bounces) {
        // (variant 13: a path's last bounce -- never its first -- needs neither the hit distance nor a next ray: pt_trace.h, pt_grid.h)
        const bool dead_end = VAR == 13 && PT_V13_DEAD_END && (n > 0) & (n == a.max_bounces - 1);
        escaped = !bounce_once<RNG, (VAR == 11 ? 11 : VAR == 13 ? 13 : 6)>(L, sc, a.n_spheres, o, d, color, mask, rng, var, n, live,
                                                                           live & prim_ok & (n == 0), dead_end);
        if (live) n++;
      }
      if (live & (escaped | (n >= a.max_bounces))) {
        if (!escaped) {
          L.color = L.color + color;                 // :198
          if (sc.lean && !(VAR == 13 && PT_V13_WELFORD_TABLE)) welford_update(var[0], luminance(color)); else welford_update(var[0], luminance(color), sc.rcpn);  // :200
        }
        i++;
        n = 0;
      }
    }
    if (by_progress) __builtin_amdgcn_s_setprio(0);
  }
  if constexpr (VAR == 12) {
    // Variant 11 with the nearest-hit search and the rest of the bounce at DIFFERENT times per lane.  In variant 11 a wave
    // walks the grid until its slowest lane is done (a ray needs 7 test trips, the wave makes 21: a quarter of the lanes
    // work) and only then shades.  Here a lane keeps its walk (GridWalk) across iterations of this loop: the wave leaves the
    // walk as soon as PT_GRID_PARK of its lanes have nothing left to do, those lanes finish their search (exact step), shade,
    // start their next ray -- next bounce, or next sample: the path regeneration of variant 10 -- and join the lanes that
    // are still walking.  Shading then runs with >= PT_GRID_PARK of 64 lanes instead of 64, the walk with most lanes busy
    // instead of a quarter.  Per pixel the sequence of operations is the reference's, as in variants 10 and 11.
    int n = 0;
    F3 o = eye, d = eye;
    F3 color = mk3(0.0f, 0.0f, 0.0f), mask = mk3(1.0f, 1.0f, 1.0f);
    GridWalk w;
    w.a = 1.0f;
    w.s = Near2{0.0f, 0.0f, 0};
    w.tmax0 = w.tmax1 = w.tmax2 = w.tdel0 = w.tdel1 = w.tdel2 = 0.0f;
    w.cidx = w.cs0 = w.cs1 = w.cs2 = 0;
    w.left = w.k0 = w.k1 = w.n0 = w.n1 = 0u;
    w.have_next = w.walking = w.forced = false;
    w.e0 = w.e1 = 0u;
    w.t_box = 0.0f;
    bool in_walk = false;  // a ray's search is under way
    bool brute = false;    // ... on the brute-force path (the grid is absent or does not admit the ray)
    for (;;) {
      if (__builtin_amdgcn_ballot_w64(i < a.spp) == 0) break;
      const bool ready = (i < a.spp) & !(in_walk & !brute & w.busy());
      if (ready) {
        if (in_walk) {  // the search of the ray in hand is over: exact step, shading (:156-196)
          float t = 0.0f;
          int idx = 0;
          bool hit = false;
          if (a.n_spheres > 0) {
            if (brute) hit = intersect_scene_screened_large(sc, a.n_spheres, o, d, make_ray_const(d), t, idx);
            else hit = grid_end(w, sc, grid, a.n_spheres, o, d, t, idx);
          }
          const bool escaped = !bounce_shade<RNG, 11>(L, sc, o, d, color, mask, rng, var, n, hit, t, idx);
          n++;
          if (escaped | (n >= a.max_bounces)) {
            if (!escaped) {
              L.color = L.color + color;                 // :198
              if (sc.lean) welford_update(var[0], luminance(color)); else welford_update(var[0], luminance(color), sc.rcpn);  // :200
            }
            i++;
            n = 0;
          }
          in_walk = false;
        }
        if (i < a.spp) {  // the next ray of this pixel
          if (n == 0) {  // :219-229
            rng.begin_sample((uint32_t)i);
            primary_ray(rng, d);
            o = eye;
            color = mk3(0.0f, 0.0f, 0.0f);
            mask = mk3(1.0f, 1.0f, 1.0f);
          }
          const float aa = dot(d, d);
          brute = !(grid.valid && a.n_spheres > 0 && grid_admits(grid, o, d, aa));
          if (!brute) grid_begin(w, grid, o, d, aa);
          in_walk = true;
        }
      }
      grid_trips<true>(w, grid, o, d, in_walk & !brute, i < a.spp, PT_GRID_PARK);
    }
  }
  if constexpr (VAR >= 7 && !kRegen && VAR != 12) {
    const int draws = (a.spp != 1 ? 2 : 0) + 2 * a.max_bounces;  // consumed by a path that never escapes
    for (; i + 2 <= a.spp; i += 2) {
      Rng<RNG> g[2] = {rng, rng};
      g[0].begin_sample((uint32_t)i);
      if constexpr (RNG == PT_RNG_XORWOW) xorwow_skip(g[1].st, draws);
      g[1].begin_sample((uint32_t)i + 1u);
      F3 o2[2] = {eye, eye}, d2[2];
      primary_ray(g[0], d2[0]);
      primary_ray(g[1], d2[1]);
      PathResult res[2];
      trace_paths<RNG, 2>(res, sc, a.n_spheres, o2, d2, g, a.max_bounces);
      accumulate_path(L, var, res[0]);
      if (RNG == PT_RNG_XORWOW && __builtin_expect(res[0].escaped, 0)) {
        // A consumed fewer draws than assumed: B's stream was wrong, retrace it from A's true state
        rng = g[0];
        rng.begin_sample((uint32_t)i + 1u);
        F3 dir;
        primary_ray(rng, dir);
        trace_ray<RNG, 6>(L, sc, a.n_spheres, eye, dir, rng, var, a.max_bounces);
      } else {
        accumulate_path(L, var, res[1]);
        rng = g[1];
      }
    }
  }
  if constexpr (!kRegen && VAR != 12) {
    // Issue priority by progress (long waves only).  A SIMD serves its waves oldest first: of four waves that start together
    // the favoured one finishes in half the time its fair share would take and the last one finishes its work alone, which a
    // lone wave cannot do at more than ~45 % of the SIMD's issue rate (a frame of 1024 workgroups x 4096 spp: first wave done
    // after 27 ms, last after 61; tools/wave_timing.py) -- and the end of EVERY kernel looks like that.  With s_setprio
    // 3, 2, 1, 0 by quarter of its own samples a wave that is ahead yields to the others on its SIMD, they reach the end
    // together, and the tail disappears (the same frame: 54 ms; the headline's four rounds: 52.4 -> 50.8 ms).  Scheduling
    // only: no value changes.  Not for short waves (the setting costs them 0.5 %; their kernels have many rounds anyway).
    const int span = i_end - i_begin;  // this workgroup's samples (all of them unless the frame is chunked)
    // ... and not for frames of many rounds of short waves.  A frame that fits the chip in ONE round (the interactive shape:
    // 512^2 x 4 spp = exactly four waves per SIMD) is all tail, however short its waves: there the priority follows the
    // sample index itself (config 5: 0.1218 -> see profiles/r03).  The launcher decides (a.prio).
    const bool by_progress = a.prio != 0u && span >= 4;
    const int q1 = i_begin + span / 4, q2 = i_begin + span / 2, q3 = i_end - span / 4;
    const int stride_mask = span >= 64 ? 15 : 0;  // how often the band is looked at
    const int i_stop = active ? i_end : i;
    for (; i < i_stop; i++) {  // :219
      if (by_progress) {
        const int iu = __builtin_amdgcn_readfirstlane(i);  // the sample index is the same in every lane that is in this loop
        if ((iu & stride_mask) == 0) {
          if (iu < q1) __builtin_amdgcn_s_setprio(3);
          else if (iu < q2) __builtin_amdgcn_s_setprio(2);
          else if (iu < q3) __builtin_amdgcn_s_setprio(1);
          else __builtin_amdgcn_s_setprio(0);
        }
      }
      rng.begin_sample((uint32_t)i);
      F3 dir;
      primary_ray(rng, dir);
      trace_ray<RNG, (VAR >= 7 ? 6 : VAR), REFB>(L, sc, a.n_spheres, eye, dir, rng, var, a.max_bounces);  // :231
    }
    if (by_progress) __builtin_amdgcn_s_setprio(0);
  }
  if constexpr (CHUNKS) {
    if (chunk + 1u < n_chunks) {  // not the last chunk: hand the pixel's state over and leave
      if (active) {
        auto st = [&](int w, uint32_t v) { a.chunk_state[(size_t)w * a.tile_pixels + tp] = v; };
        auto stf = [&](int w, float v) { st(w, __float_as_uint(v)); };
        stf(0, L.color.x); stf(1, L.color.y); stf(2, L.color.z);
        stf(3, L.normal.x); stf(4, L.normal.y); stf(5, L.normal.z);
        stf(6, L.albedo.x); stf(7, L.albedo.y); stf(8, L.albedo.z);
        stf(9, L.depth);
        st(10, (uint32_t)var[0].n);
        st(11, (uint32_t)var[1].n);
#pragma unroll
        for (int k = 0; k < 4; k++) {
          stf(12 + 2 * k, var[k].mean);
          stf(13 + 2 * k, var[k].M2);
        }
        if constexpr (RNG == PT_RNG_XORWOW) {
          st(20, rng.st.d); st(21, rng.st.v0); st(22, rng.st.v1); st(23, rng.st.v2); st(24, rng.st.v3); st(25, rng.st.v4);
        }
      }
      chunk_publish(a, block_id, chunk);
      return;
    }
  }

  const float fs = (float)a.spp;  // :234-237
  const float px[14] = {L.color.x / fs,  L.color.y / fs,  L.color.z / fs,  L.normal.x / fs, L.normal.y / fs,
                        L.normal.z / fs, L.albedo.x / fs, L.albedo.y / fs, L.albedo.z / fs, L.depth / fs,
                        welford_variance(var[0]), welford_variance(var[1]), welford_variance(var[2]),
                        welford_variance(var[3])};  // :240-254
  float* out_f = a.out;  // this frame's buffers
  float* vtx_f = a.vertices;
  if constexpr (FRAMES) {
    out_f += (size_t)fr * args.out_stride;
    if (vtx_f) vtx_f += (size_t)fr * args.vtx_stride;
  }
  if (vtx_f && active) store_display_vertex(vtx_f + (size_t)tp * 3, a.width, row, col, px[0], px[1], px[2]);
  // The 64 pixels of a wave are 64 consecutive columns, so their 64 x 14 floats are ONE contiguous
  // 3584-byte span of the [row][col][14] buffer: transpose through the wave's own LDS slice and write
  // it as 224 coalesced 16-byte stores (3.5 per lane) instead of 14 strided dword stores per lane.
  if (!REF && a.planar) {  // channel-first: consecutive lanes are consecutive columns of one plane, stores coalesce as they are
    // (the reference-configuration builds are interleaved-only: the launcher sends planar frames to the generic build)
    if (active) {
#pragma unroll
      for (int c = 0; c < 14; c++) out_f[(size_t)c * a.tile_pixels + tp] = px[c];
    }
  } else {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool wave_full = (__builtin_amdgcn_ballot_w64(active) == ~0ull) && ((reinterpret_cast<uintptr_t>(out_f) & 15u) == 0u) &&
                         VAR != 11 && VAR != 12 && VAR != 13;  // variants 11-13: the LDS holds the grid until the last wave is done: plain stores there
  if (wave_full) {
    float* wl = reinterpret_cast<float*>(lds_scene + a.scene_lds_f4) + wave * (64 * 14);
#pragma unroll
    for (int c = 0; c < 14; c++) wl[lane * 14 + c] = px[c];
    __builtin_amdgcn_wave_barrier();  // DS operations of one wave execute in order: the reads below see these writes
    const float4* src = reinterpret_cast<const float4*>(wl);
    float4* dst = reinterpret_cast<float4*>(out_f + (size_t)(tp - (uint32_t)lane) * 14);
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int q = lane + 64 * k;
      if (q < 224) dst[q] = src[q];
    }
  } else if (active) {
    float* o = out_f + (size_t)tp * 14;
#pragma unroll
    for (int c = 0; c < 14; c++) o[c] = px[c];
  }
  }
  if constexpr (FRAME_LOOP) {
    __builtin_amdgcn_wave_barrier();  // (the wave's transpose slice is reused by its next frame)
    if (++fr < fr_count) goto frame_top;
  }

  if constexpr (RNG == PT_RNG_XORWOW) {
    if (a.rng_state && active) {  // :256
      uint32_t* s = a.rng_state + (size_t)tp * 6;
      s[0] = rng.st.d;
      s[1] = rng.st.v0;
      s[2] = rng.st.v1;
      s[3] = rng.st.v2;
      s[4] = rng.st.v3;
      s[5] = rng.st.v4;
    }
  }
  if constexpr (CHUNKS) {
    if (n_chunks > 1u) chunk_publish(a, block_id, chunk);  // count == chunks: this block's frame and generator state are complete
  }
}


// ---- variant 8: four lanes per pixel (small tiles) ----------------------------------------------
// A rank of an 8-GPU run renders 131 072 pixels = 2 waves per SIMD with one lane per pixel, which is
// latency-bound (HISTORY.md B.2).  Here lane (pixel, s) traces samples s, s+4, s+8, ... so the same tile
// has 4x the waves.  What the contract fixes -- one sequential generator per pixel, sequential float
// sums, sequential Welford updates -- is preserved:
//  * xorwow: lane s starts from the pixel's state advanced by s*D draws and skips 3*D draws after each
//    of its samples, D = 2 + 2*max_bounces being what a non-escaping path consumes (speculation);
//  * after every round the four results are exchanged through the wave's LDS slice and applied IN
//    SAMPLE ORDER, feature-parallel: lane 0 owns the colour sums + colour variance, lane 1 normal,
//    lane 2 albedo, lane 3 depth -- the same additions and Welford updates as the one-lane kernel,
//    just done by four lanes side by side;
//  * if a sample that is not the frame's last one escapes (consumed fewer draws), every later
//    speculative state of that pixel is wrong: the records after it are discarded and the group
//    continues in sequential mode (lane 0 traces, all four still accumulate) from that sample's true
//    final state.  Never happens in a closed scene; in an open one the kernel degrades to 1/4
//    efficiency for that pixel but stays exact.
// Variant 9 is the same kernel with TWO lanes per pixel, each owning two features: a third of the generator
// skip-ahead (D instead of 3 D draws per sample), for tiles that are only moderately too small -- with its reference-
// configuration build (round 3) the fastest kernel on a tile of two one-lane waves per SIMD, which it turns into ONE round of
// four (the 1/8 frame of an 8-GPU run: 6.90 ms against variant 8's 7.55 and variant 6's 8.14; pt_capi.hip, small_tile_variant).
constexpr int kRecWords = 24;  // 4 feature blocks {v0,v1,v2,x} + flags + 6 state words, padded

template <int RNG, int kSplit, bool LEAN = false, int REFB = 0>
__global__ void __launch_bounds__(PT_BLOCK_THREADS) pixel_kernel_split(PixelKernelArgs a) {
  constexpr bool REF = REFB != 0;
  if constexpr (REF) {
    a.n_spheres = 9;
    a.max_bounces = REFB;
  }
  constexpr int kOwn = 4 / kSplit;  // features accumulated by one lane
  if constexpr (REF) {  // repair launch: only the pixel blocks a broken chunked launch left untouched (pixel_kernel)
    if (a.repair != 0u && __hip_atomic_load(a.chunk_flag + blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a.repair) return;
  }
  extern __shared__ float4 lds_scene[];
  SceneLds sc = stage_scene<false>(a.spheres, a.n_spheres, lds_scene, LEAN, mk3(a.eye[0], a.eye[1], a.eye[2]), a.spp);
  sc.small_only = !LEAN && (kSplit == 4 || REF);  // variant 8 has a lean build for larger scenes, variant 9 has not (its REF builds see 9 spheres)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* xl = reinterpret_cast<float*>(lds_scene + a.scene_lds_f4) + wave * (64 * kRecWords);
  const int gbase = lane & ~(kSplit - 1);

  // Sample chunking as in pixel_kernel (reference-configuration builds): workgroup blockIdx.x = chunk * n_blocks + block renders
  // the samples [chunk * per, (chunk + 1) * per) of its 256 / kSplit pixels and hands their state to the next chunk through
  // chunk_state (same 26-word layout: every lane stores the features it owns, lane 0 of a pixel the true generator state).
  constexpr bool CHUNKS = REF;
  uint32_t block_id = blockIdx.x, chunk = 0u, n_chunks = 1u;
  if constexpr (CHUNKS) {
    if (a.chunks > 1u) {
      const uint32_t n_blocks = (uint32_t)(((uint64_t)a.tile_pixels * kSplit + PT_BLOCK_THREADS - 1) / PT_BLOCK_THREADS);
      n_chunks = a.chunks;
      chunk = blockIdx.x / n_blocks;
      block_id = blockIdx.x - chunk * n_blocks;
    }
  }
  const uint32_t gl = block_id * PT_BLOCK_THREADS + threadIdx.x;
  const uint32_t tp = gl / kSplit;      // pixel index inside the tile
  const int s = (int)(gl % kSplit);     // sample slot; this lane accumulates features s*kOwn .. s*kOwn + kOwn - 1
  const bool active = tp < a.tile_pixels;
  const int row = a.row_begin + (int)(tp / (uint32_t)a.width);
  const int col = (int)(tp % (uint32_t)a.width);
  const uint32_t id = (uint32_t)row * (uint32_t)a.width + (uint32_t)col;

  Rng<RNG> rng;
  const int D = (a.spp != 1 ? 2 : 0) + 2 * a.max_bounces;
  if constexpr (RNG == PT_RNG_XORWOW) {
    if (a.rng_state && active) {
      const uint32_t* p = a.rng_state + (size_t)tp * 6;
      rng.st = Xorwow{p[0], p[1], p[2], p[3], p[4], p[5]};
    } else {
      xorwow_init(rng.st, (uint64_t)id + a.seed);
    }
    xorwow_skip(rng.st, s * D);
  } else {
    rng.k0 = (uint32_t)a.seed;
    rng.k1 = (uint32_t)(a.seed >> 32) ^ a.frame;
    rng.pix = id;
  }
  Xorwow final_state = Xorwow{0, 0, 0, 0, 0, 0};
  if constexpr (RNG == PT_RNG_XORWOW) final_state = rng.st;  // spp == 0 never happens; overwritten below

  const F3 B0 = mk3(a.basis[0], a.basis[1], a.basis[2]), B1 = mk3(a.basis[3], a.basis[4], a.basis[5]);
  const F3 B2 = mk3(a.basis[6], a.basis[7], a.basis[8]), B3 = mk3(a.basis[9], a.basis[10], a.basis[11]);
  const F3 eye = mk3(a.eye[0], a.eye[1], a.eye[2]);
  const bool pow2_image = ((a.width & (a.width - 1)) == 0) && ((a.height & (a.height - 1)) == 0);  // wave-uniform
  const float inv_w = 1.0f / (float)a.width, inv_h = 1.0f / (float)a.height;  // exact for powers of two

  if constexpr (REF) {
#ifndef PT_NO_FOOTPRINT
    if (a.spp >= PT_FOOTPRINT_MIN_SPP) {  // once per pixel: the spheres its primary rays can return (pt_footprint.h); the wave ranks the union
      auto dir_at = [&](float sx, float sy) {
        if (pow2_image) {
          sx *= inv_h;
          sy *= inv_w;
        } else {
          sx /= (float)a.height;
          sy /= (float)a.width;
        }
        return lerp(lerp(B0, B1, sy), lerp(B2, B3, sy), 1.0f - sx);
      };
      const uint32_t mine = active ? primary_candidates(sc, a.n_spheres, (float)row, (float)col, dir_at) : 0u;
      uint32_t wave_mask = 0u;
#pragma unroll
      for (int j = 0; j < 9; j++) wave_mask |= (__builtin_amdgcn_ballot_w64(((mine >> j) & 1u) != 0u) != 0ull) ? (1u << j) : 0u;
      sc.prim_mask = wave_mask;
    }
#endif
  }

  float sum0[kOwn], sum1[kOwn], sum2[kOwn];  // this lane's feature sums (depth uses sum0 only)
  Welford w[kOwn];
#pragma unroll
  for (int q = 0; q < kOwn; q++) {
    sum0[q] = sum1[q] = sum2[q] = 0.0f;
    w[q] = Welford{0, 0.0f, 0.0f};
  }
  bool seq = false;  // group-uniform: sequential mode after a failed speculation
  int i_begin = 0, i_end = a.spp;  // this workgroup's samples (all of them unless the frame is chunked)
  if constexpr (CHUNKS) {
    if (n_chunks > 1u) {
      int per = (a.spp + (int)n_chunks - 1) / (int)n_chunks;
      per = (per + kSplit - 1) / kSplit * kSplit;  // whole rounds
      i_begin = (int)chunk * per < a.spp ? (int)chunk * per : a.spp;
      i_end = i_begin + per < a.spp ? i_begin + per : a.spp;
      if (chunk > 0u) {
        if (chunk_wait(a, block_id, chunk)) return;  // broken chain: nothing of the block is written (pixel_kernel)
        if (active) {
          auto ld = [&](int w) { return __hip_atomic_load(a.chunk_state + (size_t)w * a.tile_pixels + tp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
          auto ldf = [&](int w) { return __uint_as_float(ld(w)); };
#pragma unroll
          for (int q = 0; q < kOwn; q++) {
            const int f = s * kOwn + q;  // 0 colour, 1 normal, 2 albedo, 3 depth
            sum0[q] = ldf(f < 3 ? 3 * f : 9);
            sum1[q] = f < 3 ? ldf(3 * f + 1) : 0.0f;
            sum2[q] = f < 3 ? ldf(3 * f + 2) : 0.0f;
            w[q] = Welford{(int)ld(f == 0 ? 10 : 11), ldf(12 + 2 * f), ldf(13 + 2 * f)};
          }
          if constexpr (RNG == PT_RNG_XORWOW) {  // the pixel's true state after sample i_begin - 1; lane s starts s samples further on
            rng.st = Xorwow{ld(20), ld(21), ld(22), ld(23), ld(24), ld(25)};
            final_state = rng.st;
            xorwow_skip(rng.st, s * D);
          }
        }
      }
    }
  }
  int base = i_begin;  // group-uniform: first sample of the current round

  // issue priority by progress, as in pixel_kernel (there: why); the progress of a wave is that of its first pixel group
  const int span = i_end - i_begin;
  const bool by_progress = a.spp >= PT_PRIO_MIN_SPP && span >= 4;
  const int q1 = i_begin + span / 4, q2 = i_begin + span / 2, q3 = i_end - span / 4;
  int last_band = -1;
  while (base < i_end) {
    if (by_progress) {
      const int bu = __builtin_amdgcn_readfirstlane(base);
      const int band = (bu >= q1 ? 1 : 0) + (bu >= q2 ? 1 : 0) + (bu >= q3 ? 1 : 0);
      if (band != last_band) {  // wave-uniform
        last_band = band;
        if (band == 0) __builtin_amdgcn_s_setprio(3);
        else if (band == 1) __builtin_amdgcn_s_setprio(2);
        else if (band == 2) __builtin_amdgcn_s_setprio(1);
        else __builtin_amdgcn_s_setprio(0);
      }
    }
    const int k = seq ? base : base + s;
    const bool mine = active && (!seq || s == 0) && k < i_end;
    PathResult res[1];
    res[0].color = res[0].normal0 = res[0].albedo0 = mk3(0.0f, 0.0f, 0.0f);
    res[0].t0 = 0.0f;
    res[0].hit0 = res[0].escaped = false;
    if (mine) {
      Rng<RNG> g[1] = {rng};
      g[0].begin_sample((uint32_t)k);
      float sx = (float)row, sy = (float)col;  // :221-229
      if (a.spp != 1) {
        float jx, jy;
        g[0].jitter(jx, jy);
        sx += jx * 1.0f - 0.5f;
        sy += jy * 1.0f - 0.5f;
      }
      if (pow2_image) {
        sx *= inv_h;
        sy *= inv_w;
      } else {
        sx /= (float)a.height;
        sy /= (float)a.width;
      }
      F3 o1[1] = {eye}, d1[1] = {lerp(lerp(B0, B1, sy), lerp(B2, B3, sy), 1.0f - sx)};
      trace_paths<RNG, 1, REFB>(res, sc, a.n_spheres, o1, d1, g, a.max_bounces);
      rng = g[0];
    }
    // publish this lane's sample
    float* rec = xl + lane * kRecWords;
    const float lumC = luminance(res[0].color), lumN = luminance(res[0].normal0);
    const float lumA = sc.lean ? luminance(res[0].albedo0) : res[0].lum_albedo0;  // staged per sphere where the scene image is (pt_scene_lds.h)
    const uint32_t own_fl = (mine ? 1u : 0u) | (res[0].hit0 ? 2u : 0u) | (res[0].escaped ? 4u : 0u);
    *reinterpret_cast<float4*>(rec + 0) = make_float4(res[0].color.x, res[0].color.y, res[0].color.z, lumC);
    *reinterpret_cast<float4*>(rec + 4) = make_float4(res[0].normal0.x, res[0].normal0.y, res[0].normal0.z, lumN);
    *reinterpret_cast<float4*>(rec + 8) = make_float4(res[0].albedo0.x, res[0].albedo0.y, res[0].albedo0.z, lumA);
    *reinterpret_cast<float4*>(rec + 12) = make_float4(res[0].t0, 0.0f, 0.0f, res[0].t0);
    reinterpret_cast<uint32_t*>(rec)[16] = own_fl;
    if constexpr (RNG == PT_RNG_XORWOW) {
      uint32_t* ru = reinterpret_cast<uint32_t*>(rec) + 17;
      ru[0] = rng.st.d; ru[1] = rng.st.v0; ru[2] = rng.st.v1; ru[3] = rng.st.v2; ru[4] = rng.st.v3; ru[5] = rng.st.v4;
    }
    __builtin_amdgcn_wave_barrier();
    // apply the round's samples in order to this lane's feature
    bool failed = false;
    int consumed = 0;  // samples of this round that count (all of them unless a speculation failed)
    // Every record of every pixel group of this wave a traced sample that hit something and stayed inside (wave-uniform; in a
    // closed scene: always, except in a ragged last wave): every enable below is true, so the sums and the Welford updates
    // run unconditionally -- no selects -- and the generator state is the last record's.  Same operations in the same order.
    const bool all_hits = __builtin_amdgcn_ballot_w64(own_fl != 3u) == 0;
    if (all_hits) {
#pragma unroll
      for (int j = 0; j < kSplit; j++) {
        const float* rj = xl + (gbase + j) * kRecWords;
#pragma unroll
        for (int q = 0; q < kOwn; q++) {
          const float4 v = *reinterpret_cast<const float4*>(rj + 4 * (s * kOwn + q));
          sum0[q] = sum0[q] + v.x;
          sum1[q] = sum1[q] + v.y;
          sum2[q] = sum2[q] + v.z;
          if (sc.lean) welford_update(w[q], v.w); else welford_update(w[q], v.w, sc.rcpn);
        }
      }
      if constexpr (RNG == PT_RNG_XORWOW) {
        const uint32_t* ru = reinterpret_cast<const uint32_t*>(xl + (gbase + kSplit - 1) * kRecWords) + 17;
        final_state = Xorwow{ru[0], ru[1], ru[2], ru[3], ru[4], ru[5]};
      }
      consumed = kSplit;
    } else {
#pragma unroll
    for (int j = 0; j < kSplit; j++) {
      const float* rj = xl + (gbase + j) * kRecWords;
      const uint32_t fl = reinterpret_cast<const uint32_t*>(rj)[16];
      const bool valid = ((fl & 1u) != 0u) & !failed;
      const bool hit0 = (fl & 2u) != 0u, esc = (fl & 4u) != 0u;
#pragma unroll
      for (int q = 0; q < kOwn; q++) {
        const int f = s * kOwn + q;  // 0 colour, 1 normal, 2 albedo, 3 depth
        const float4 v = *reinterpret_cast<const float4*>(rj + 4 * f);
        const bool en_sum = valid & ((f == 0) | hit0);        // colour is always added (:159,:198); AOVs on a first hit (:187-190)
        const bool en_w = valid & ((f == 0) ? !esc : hit0);    // :200 / :192-194
        sum0[q] = en_sum ? sum0[q] + v.x : sum0[q];
        sum1[q] = en_sum ? sum1[q] + v.y : sum1[q];
        sum2[q] = en_sum ? sum2[q] + v.z : sum2[q];
        Welford wn = w[q];
        if (sc.lean) welford_update(wn, v.w); else welford_update(wn, v.w, sc.rcpn);
        w[q].n = en_w ? wn.n : w[q].n;  // field-wise: a struct select would go through scratch
        w[q].mean = en_w ? wn.mean : w[q].mean;
        w[q].M2 = en_w ? wn.M2 : w[q].M2;
      }
      if constexpr (RNG == PT_RNG_XORWOW) {
        if (valid) {
          const uint32_t* ru = reinterpret_cast<const uint32_t*>(rj) + 17;
          final_state = Xorwow{ru[0], ru[1], ru[2], ru[3], ru[4], ru[5]};
        }
        const int kj = seq ? base : base + j;
        if (valid & esc & !seq & (kj < i_end - 1)) failed = true;  // later speculative states are wrong
      }
      consumed += valid ? 1 : 0;
    }
    }
    __builtin_amdgcn_wave_barrier();  // records are rewritten next round
    if (!seq) {
      if (failed) {
        if (s == 0 && a.fail_count) atomicAdd(a.fail_count, 1u);  // lets the host stop speculating on open scenes
        seq = true;
        base += consumed;          // resume right after the escaped sample
        if constexpr (RNG == PT_RNG_XORWOW) rng.st = final_state;  // its true final state (only lane 0 traces from here on)
      } else {
        base += kSplit;
        if constexpr (RNG == PT_RNG_XORWOW) {
          // (reference-configuration builds: the step count is a compile-time constant unless the frame has one sample per pixel)
          if (REFB != 0 && a.spp != 1) xorwow_skip_n<(kSplit - 1) * (2 + 2 * (REFB != 0 ? REFB : 1))>(rng.st);
          else xorwow_skip(rng.st, (kSplit - 1) * D);
        }
      }
    } else {
      base += 1;
      if constexpr (RNG == PT_RNG_XORWOW) rng.st = final_state;
    }
  }
  if (by_progress) __builtin_amdgcn_s_setprio(0);

  if constexpr (CHUNKS) {
    if (chunk + 1u < n_chunks) {  // not the last chunk: hand the pixel's state over and leave
      if (active) {
        auto st = [&](int wd, uint32_t v) { a.chunk_state[(size_t)wd * a.tile_pixels + tp] = v; };
        auto stf = [&](int wd, float v) { st(wd, __float_as_uint(v)); };
#pragma unroll
        for (int q = 0; q < kOwn; q++) {
          const int f = s * kOwn + q;
          if (f < 3) {
            stf(3 * f, sum0[q]); stf(3 * f + 1, sum1[q]); stf(3 * f + 2, sum2[q]);
          } else {
            stf(9, sum0[q]);
          }
          if (f < 2) st(10 + f, (uint32_t)w[q].n);  // the three first-hit accumulators count together: the normal's stands for them
          stf(12 + 2 * f, w[q].mean);
          stf(13 + 2 * f, w[q].M2);
        }
        if constexpr (RNG == PT_RNG_XORWOW) {
          if (s == 0) {
            st(20, final_state.d); st(21, final_state.v0); st(22, final_state.v1); st(23, final_state.v2); st(24, final_state.v3); st(25, final_state.v4);
          }
        }
      }
      chunk_publish(a, block_id, chunk);
      return;
    }
  }

  if (active) {  // :234-254, each lane stores its feature
    const float fs = (float)a.spp;
    // channel c of this pixel: interleaved [row][col][14] or channel-first [14][tile pixels]
    const bool planar = !REF && a.planar;
    float* o = planar ? a.out + tp : a.out + (size_t)tp * 14;
    const size_t cs = planar ? (size_t)a.tile_pixels : 1;
#pragma unroll
    for (int q = 0; q < kOwn; q++) {
      const int f = s * kOwn + q;
      if (f < 3) {
        const float v0 = sum0[q] / fs, v1 = sum1[q] / fs, v2 = sum2[q] / fs;
        o[(3 * f + 0) * cs] = v0;
        o[(3 * f + 1) * cs] = v1;
        o[(3 * f + 2) * cs] = v2;
        if (f == 0 && a.vertices) store_display_vertex(a.vertices + (size_t)tp * 3, a.width, row, col, v0, v1, v2);  // the colour's owner
      } else {
        o[9 * cs] = sum0[q] / fs;
      }
      o[(10 + f) * cs] = welford_variance(w[q]);
    }
    if constexpr (RNG == PT_RNG_XORWOW) {
      if (a.rng_state && s == 0) {  // :256
        uint32_t* p = a.rng_state + (size_t)tp * 6;
        p[0] = final_state.d; p[1] = final_state.v0; p[2] = final_state.v1;
        p[3] = final_state.v2; p[4] = final_state.v3; p[5] = final_state.v4;
      }
    }
  }
  if constexpr (CHUNKS) {
    if (n_chunks > 1u) chunk_publish(a, block_id, chunk);  // count == chunks: this block is complete
  }
}

// setup_random: src/pathtrace.cu:259-266
__global__ void __launch_bounds__(PT_BLOCK_THREADS)
    setup_random_kernel(uint32_t* state, int width, int row_begin, uint32_t tile_pixels, uint64_t seed) {
  const uint32_t tp = blockIdx.x * PT_BLOCK_THREADS + threadIdx.x;
  if (tp >= tile_pixels) return;
  const uint32_t id = (uint32_t)(row_begin + (int)(tp / (uint32_t)width)) * (uint32_t)width + tp % (uint32_t)width;
  Xorwow s;
  xorwow_init(s, (uint64_t)id + seed);
  uint32_t* p = state + (size_t)tp * 6;
  p[0] = s.d;
  p[1] = s.v0;
  p[2] = s.v1;
  p[3] = s.v2;
  p[4] = s.v3;
  p[5] = s.v4;
}

}  // namespace pt

// ---- launchers (host) ---------------------------------------------------------------------
// LDS layout of a launch (pt_scene_lds.h): many-sphere scenes keep only the geometry in LDS
// (variants 6, 8 and 10 -- the ones the automatic policy uses -- are also built for that layout)
static inline bool lds_lean(int n, int variant) {
  return variant == 11 || variant == 12 || variant == 13 || variant == 14 || (n > PT_SCREEN_MAX_SPHERES && (variant == 6 || variant == 8 || variant == 10));
}
static inline bool is_split(int variant) { return variant == 8 || variant == 9; }
static inline size_t scene_lds_f4(int n, int variant) {
  if (lds_lean(n, variant)) return pt::kTablesF4;  // the lean builds read the caller's array directly: only the small tables
  return (size_t)n * 4 + pt::kTablesF4 + (variant == 3 ? (size_t)((n + 1) / 2) * 2 : 0);  // geometry, two material slots, the eye image, the unit-length table
}
// what follows the scene image: one 64 x 14 float transpose slice per wave for the epilogue, or the
// split kernels' exchange records
static inline size_t tail_lds_bytes(int n, int variant) {
  if (variant == 11 || variant == 12) return n <= pt::kGridMaxSpheres ? pt::grid_lds_bytes(n, false) : 64;  // geometry + grid tables instead of the epilogue slice
  if (variant == 13 || variant == 14) {  // + per wave the test ring and the owners' result slots
    const int threads = variant == 14 ? PT_GRID_WIDE_THREADS : PT_GRID_BLOCK_THREADS;
    return (n <= pt::kGridMaxSpheres ? ((pt::grid_lds_bytes(n, true, threads) + 15) & ~(size_t)15) : 64) + (threads / 64) * pt::kPoolWaveBytes;
  }
  if (is_split(variant)) return (PT_BLOCK_THREADS / 64) * 64 * pt::kRecWords * sizeof(float);
  return (PT_BLOCK_THREADS / 64) * 64 * 14 * sizeof(float);
}
static inline size_t scene_lds_bytes(int n, int variant) { return scene_lds_f4(n, variant) * sizeof(float4) + tail_lds_bytes(n, variant); }

typedef void (*pixel_kernel_fn)(PixelKernelArgs);

// the bounce cap of the reference-configuration build these launch parameters run (5 or 8), 0 = a generic build
static inline int ref_config(int n, int max_bounces, int variant, bool planar) {
  return (n == 9 && (max_bounces == 5 || max_bounces == 8) && !planar && (variant == 6 || variant == 8 || variant == 9)) ? max_bounces : 0;
}
int pt_kernel_ref_bounces(int n_spheres, int max_bounces, int variant, bool planar) { return ref_config(n_spheres, max_bounces, variant, planar); }

static pixel_kernel_fn select_kernel(int rng_mode, int variant, bool lean, int ref) {
  const bool philox = rng_mode == PT_RNG_PHILOX;
  if (ref == 5 && !lean) {
    if (variant == 6) return philox ? pt::pixel_kernel<PT_RNG_PHILOX, 6, false, 5> : pt::pixel_kernel<PT_RNG_XORWOW, 6, false, 5>;
    if (variant == 8)
      return philox ? pt::pixel_kernel_split<PT_RNG_PHILOX, 4, false, 5> : pt::pixel_kernel_split<PT_RNG_XORWOW, 4, false, 5>;
    if (variant == 9)
      return philox ? pt::pixel_kernel_split<PT_RNG_PHILOX, 2, false, 5> : pt::pixel_kernel_split<PT_RNG_XORWOW, 2, false, 5>;
  }
  if (ref == 8 && !lean) {
    if (variant == 6) return philox ? pt::pixel_kernel<PT_RNG_PHILOX, 6, false, 8> : pt::pixel_kernel<PT_RNG_XORWOW, 6, false, 8>;
    if (variant == 8)
      return philox ? pt::pixel_kernel_split<PT_RNG_PHILOX, 4, false, 8> : pt::pixel_kernel_split<PT_RNG_XORWOW, 4, false, 8>;
    if (variant == 9)
      return philox ? pt::pixel_kernel_split<PT_RNG_PHILOX, 2, false, 8> : pt::pixel_kernel_split<PT_RNG_XORWOW, 2, false, 8>;
  }
  if (lean) {
    switch (variant) {
      case 6: return philox ? pt::pixel_kernel<PT_RNG_PHILOX, 6, true> : pt::pixel_kernel<PT_RNG_XORWOW, 6, true>;
      case 8: return philox ? pt::pixel_kernel_split<PT_RNG_PHILOX, 4, true> : pt::pixel_kernel_split<PT_RNG_XORWOW, 4, true>;
      case 10: return philox ? pt::pixel_kernel<PT_RNG_PHILOX, 10, true> : pt::pixel_kernel<PT_RNG_XORWOW, 10, true>;
      case 13: return philox ? pt::pixel_kernel<PT_RNG_PHILOX, 13, true> : pt::pixel_kernel<PT_RNG_XORWOW, 13, true>;
      case 14: return philox ? pt::pixel_kernel<PT_RNG_PHILOX, 13, true, 0, false, true> : pt::pixel_kernel<PT_RNG_XORWOW, 13, true, 0, false, true>;
#if PT_BUILD_EXPERIMENTS  // variant 11 (the grid walk with every lane testing its own spheres): superseded by 13, kept for A/B
      case 11: return philox ? pt::pixel_kernel<PT_RNG_PHILOX, 11, true> : pt::pixel_kernel<PT_RNG_XORWOW, 11, true>;
      case 12: return philox ? pt::pixel_kernel<PT_RNG_PHILOX, 12, true> : pt::pixel_kernel<PT_RNG_XORWOW, 12, true>;
#endif
      default: return nullptr;
    }
  }
  switch (variant) {
    case 0: return philox ? pt::pixel_kernel<PT_RNG_PHILOX, 0> : pt::pixel_kernel<PT_RNG_XORWOW, 0>;
    case 9: return philox ? pt::pixel_kernel_split<PT_RNG_PHILOX, 2> : pt::pixel_kernel_split<PT_RNG_XORWOW, 2>;
#if PT_BUILD_EXPERIMENTS  // measured negative results and stepping stones: libptcore_lab.so only
    case 1: return philox ? pt::pixel_kernel<PT_RNG_PHILOX, 1> : pt::pixel_kernel<PT_RNG_XORWOW, 1>;
    case 2: return philox ? pt::pixel_kernel<PT_RNG_PHILOX, 2> : pt::pixel_kernel<PT_RNG_XORWOW, 2>;
    case 3: return philox ? pt::pixel_kernel<PT_RNG_PHILOX, 3> : pt::pixel_kernel<PT_RNG_XORWOW, 3>;
    case 4: return philox ? pt::pixel_kernel<PT_RNG_PHILOX, 4> : pt::pixel_kernel<PT_RNG_XORWOW, 4>;
    case 5: return philox ? pt::pixel_kernel<PT_RNG_PHILOX, 5> : pt::pixel_kernel<PT_RNG_XORWOW, 5>;
    case 7: return philox ? pt::pixel_kernel<PT_RNG_PHILOX, 7> : pt::pixel_kernel<PT_RNG_XORWOW, 7>;
#endif
    case 6: return philox ? pt::pixel_kernel<PT_RNG_PHILOX, 6> : pt::pixel_kernel<PT_RNG_XORWOW, 6>;
    case 8: return philox ? pt::pixel_kernel_split<PT_RNG_PHILOX, 4> : pt::pixel_kernel_split<PT_RNG_XORWOW, 4>;
    case 10: return philox ? pt::pixel_kernel<PT_RNG_PHILOX, 10> : pt::pixel_kernel<PT_RNG_XORWOW, 10>;
    default: return nullptr;
  }
}

#ifdef PT_SCREEN_STATS
extern "C" int pt_debug_screen_stats(unsigned long long out[8], int reset) {
  unsigned long long zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(pt::g_screen_stats), sizeof(zero)) != hipSuccess) return -2;
  if (reset && hipMemcpyToSymbol(HIP_SYMBOL(pt::g_screen_stats), zero, sizeof(zero)) != hipSuccess) return -2;
  return 0;
}
#endif

#ifdef PT_GRID_STATS
extern "C" int pt_debug_grid_hist(unsigned long long out[256], int reset) {
  static unsigned long long zero[256];
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(pt::g_grid_hist), sizeof(zero)) != hipSuccess) return -2;
  if (reset && hipMemcpyToSymbol(HIP_SYMBOL(pt::g_grid_hist), zero, sizeof(zero)) != hipSuccess) return -2;
  return 0;
}
extern "C" int pt_debug_grid_stats(unsigned long long out[8], int reset) {
  unsigned long long zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(pt::g_grid_stats), sizeof(zero)) != hipSuccess) return -2;
  if (reset && hipMemcpyToSymbol(HIP_SYMBOL(pt::g_grid_stats), zero, sizeof(zero)) != hipSuccess) return -2;
  return 0;
}
#endif

int pt_kernel_num_variants(void) { return 15; }

int pt_kernel_block_threads(int variant) {
  return variant == 14 ? PT_GRID_WIDE_THREADS : (variant == 11 || variant == 12 || variant == 13) ? PT_GRID_BLOCK_THREADS : PT_BLOCK_THREADS;
}

bool pt_kernel_has_variant(int variant) {
  if (variant < 0 || variant >= 15) return false;
#if PT_BUILD_EXPERIMENTS
  return true;
#else
  return variant == 0 || variant == 6 || variant == 8 || variant == 9 || variant == 10 || variant == 13 || variant == 14;
#endif
}

size_t pt_kernel_accel_bytes(void) { return pt::kGridAccelBytes; }


const void* pt_kernel_symbol(int rng_mode, int variant, int n_spheres, int max_bounces, bool planar) {
  if (variant == 12 && max_bounces < 1) variant = 11;
  return (const void*)select_kernel(rng_mode, variant, lds_lean(n_spheres, variant), ref_config(n_spheres, max_bounces, variant, planar));
}

size_t pt_kernel_lds_bytes(int n_spheres, int variant) { return scene_lds_bytes(n_spheres, variant); }

int pt_kernel_max_spheres(int variant) {
  // variants with a lean build stage nothing for big scenes; the others are bounded by their LDS image
  const size_t tail = tail_lds_bytes(0, variant);
  if (variant == 6 || variant == 8 || variant == 10 || variant == 11 || variant == 12 || variant == 13 || variant == 14) return 1 << 26;  // byte offsets of the 40-byte records stay inside 32 bits
  const size_t fixed = tail + pt::kTablesF4 * sizeof(float4);
  if (variant == 3) return (int)((PT_LDS_BUDGET_BYTES - fixed - 2 * sizeof(float4)) / (5 * sizeof(float4)));
  return (int)((PT_LDS_BUDGET_BYTES - fixed) / (4 * sizeof(float4)));
}

hipError_t pt_launch_build_grid(const pt_sphere* spheres, int n, uint32_t* accel, const float* eye, bool pooled, hipStream_t stream, int threads) {
  // (table entries the LDS image of the kernel that will walk this grid has room for)
  hipLaunchKernelGGL(pt::build_grid_kernel, dim3(1), dim3(pt::kGridBuildThreads), 0, stream, spheres, n, accel, eye ? eye[0] : 0.0f,
                     eye ? eye[1] : 0.0f, eye ? eye[2] : 0.0f, eye ? 1 : 0, pt::grid_max_entries(n, pooled, threads));
  return hipGetLastError();
}

// where the parts of the builder's image lie (lab diagnostics: pt_debug_grid_image)
void pt_kernel_grid_layout(int n_spheres, int threads, uint64_t out[8]) {
  out[0] = pt::kGridAccelBytes;
  out[1] = pt::kGridBigOff;
  out[2] = pt::kGridStartOff;
  out[3] = pt::kGridItemsOff;
  out[4] = pt::kGridCellsOff;
  out[5] = pt::kGridEmisOff;
  out[6] = (uint64_t)pt::kGridMaxCells;
  out[7] = (uint64_t)pt::grid_max_entries(n_spheres, true, threads);
}

// does a launch with these arguments chain a pixel's samples through several workgroups (sample chunking)?
bool pt_kernel_chunked(int variant, int n_spheres, int max_bounces, bool planar, int spp, uint32_t chunks) {
  return (((variant == 6 || variant == 8 || variant == 9) && !lds_lean(n_spheres, variant) && ref_config(n_spheres, max_bounces, variant, planar)) || variant == 13) && chunks > 1u &&
         chunks <= (uint32_t)PT_CHUNKS_MAX && spp >= 2 * (int)chunks &&
         (spp + (int)chunks - 1) / (int)chunks <= ((variant == 13 || variant == 14) ? PT_CHUNK_MAX_SAMPLES_GRID : PT_CHUNK_MAX_SAMPLES);
}

hipError_t pt_launch_pixel_kernel(const PixelKernelArgs& a, int rng_mode, int variant, hipStream_t stream) {
  if (variant == 12 && a.max_bounces < 1) variant = 11;  // variant 12's loop assumes every ray is searched for
  pixel_kernel_fn fn = select_kernel(rng_mode, variant, lds_lean(a.n_spheres, variant), ref_config(a.n_spheres, a.max_bounces, variant, a.planar != 0u));
  if (!fn) return hipErrorInvalidValue;
  PixelKernelArgs b = a;
  b.scene_lds_f4 = (uint32_t)scene_lds_f4(a.n_spheres, variant);
  b.prio = (a.spp >= PT_PRIO_MIN_SPP || a.prio != 0u) ? 1u : 0u;  // long waves always; short ones when the caller says the frame is one round
  const size_t lds = scene_lds_bytes(a.n_spheres, variant);
  const size_t lds_budget = variant == 14 ? (size_t)PT_LDS_WIDE_BUDGET_BYTES : (size_t)PT_LDS_BUDGET_BYTES;
  if (lds > lds_budget) return hipErrorInvalidValue;
  if (lds > 64 * 1024) {  // beyond the default dynamic-LDS limit: opt in (gfx950 has 160 KiB per CU)
    hipError_t e = hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_budget);
    if (e != hipSuccess) return e;
  }
  if (variant == 11 || variant == 12 || variant == 13 || variant == 14) {  // this frame's grid (the scene may have changed since the last one)
    if (!a.accel) return hipErrorInvalidValue;
    hipError_t e = pt_launch_build_grid(a.spheres, a.n_spheres, const_cast<uint32_t*>(a.accel), a.eye, variant == 13 || variant == 14, stream,
                                        pt_kernel_block_threads(variant));
    if (e != hipSuccess) return e;
  }
  const uint64_t lanes = (uint64_t)a.tile_pixels * (uint64_t)(variant == 8 ? 4 : variant == 9 ? 2 : 1);
  const unsigned block = (unsigned)pt_kernel_block_threads(variant);
  unsigned grid = (unsigned)((lanes + block - 1) / block);
  // sample chunking: only the reference-configuration builds of variant 6 and variant 13 hand a pixel's state from workgroup to workgroup
  const bool chunked = a.repair == 0u && pt_kernel_chunked(variant, a.n_spheres, a.max_bounces, a.planar != 0u, a.spp, a.chunks) && a.chunk_state && a.chunk_flag;
  if (a.repair != 0u && !a.chunk_flag) return hipErrorInvalidValue;  // a repair launch reads the flags its chunked predecessor left
  b.chunks = chunked ? a.chunks : 0u;
  if (chunked) {
    hipError_t e = hipMemsetAsync(a.chunk_flag, 0, (size_t)grid * sizeof(uint32_t), stream);
    if (e != hipSuccess) return e;
    grid *= a.chunks;
  }
  hipLaunchKernelGGL(fn, dim3(grid), dim3(block), lds, stream, b);
  return hipGetLastError();
}

bool pt_kernel_has_frames(int variant, int n_spheres, int max_bounces, bool planar) {
  return variant == 6 && ref_config(n_spheres, max_bounces, variant, planar) != 0;
}

// XORWOW: one workgroup per pixel block, looping over the batch's frames; philox: one per (frame, pixel block) (pixel_kernel, FRAMES)
hipError_t pt_launch_frames_kernel(const FramesKernelArgs& fa, int rng_mode, hipStream_t stream) {
  const PixelKernelArgs& a = fa.base;
  const int ref = ref_config(a.n_spheres, a.max_bounces, 6, a.planar != 0u);
  if (ref == 0 || fa.frames < 2u || fa.frames > (uint32_t)PT_FRAMES_MAX) return hipErrorInvalidValue;
  const bool philox = rng_mode == PT_RNG_PHILOX;
  typedef void (*frames_fn)(FramesKernelArgs);
  frames_fn fn = ref == 5 ? (philox ? pt::pixel_kernel<PT_RNG_PHILOX, 6, false, 5, true> : pt::pixel_kernel<PT_RNG_XORWOW, 6, false, 5, true>)
                          : (philox ? pt::pixel_kernel<PT_RNG_PHILOX, 6, false, 8, true> : pt::pixel_kernel<PT_RNG_XORWOW, 6, false, 8, true>);
  FramesKernelArgs b = fa;
  b.base.scene_lds_f4 = (uint32_t)scene_lds_f4(a.n_spheres, 6);
  b.base.prio = (a.spp >= PT_PRIO_MIN_SPP || a.prio != 0u) ? 1u : 0u;
  b.base.chunks = 0u;
  b.base.repair = 0u;
  const size_t lds = scene_lds_bytes(a.n_spheres, 6);
  const unsigned block = (unsigned)PT_BLOCK_THREADS;
  const unsigned blocks = (unsigned)(((uint64_t)a.tile_pixels + block - 1) / block);
  hipLaunchKernelGGL(fn, dim3(philox ? blocks * fa.frames : blocks), dim3(block), lds, stream, b);
  return hipGetLastError();
}

hipError_t pt_launch_setup_random(uint32_t* state, int width, int row_begin, uint32_t tile_pixels, uint64_t seed,
                                  hipStream_t stream) {
  const unsigned grid = (tile_pixels + PT_BLOCK_THREADS - 1) / PT_BLOCK_THREADS;
  hipLaunchKernelGGL(pt::setup_random_kernel, dim3(grid), dim3(PT_BLOCK_THREADS), 0, stream, state, width, row_begin,
                     tile_pixels, seed);
  return hipGetLastError();
}
