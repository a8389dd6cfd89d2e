/*
 * ptcore.h -- C ABI of libptcore.so: the MI355X (gfx950) implementation of the per-pixel
 * Monte-Carlo path-trace megakernel of trevor-m/cuda-pathtrace.
 *
 * This is the drop-in boundary.  The reference has no FFI layer; its boundary for this path
 * is header-level C++ (include/pathtrace.h, include/Renderer.h, include/OutputBuffer.h,
 * include/Scene.h).  Every entry point below names the reference interface it replaces
 * (file:line under the reference tree).  Plain pointers and sizes only; no C++ or torch
 * types.  The C++ look-alike classes that forward to these functions live in
 * cuda-pathtrace_amd/host/ (Renderer.h, OutputBuffer.h, Scene.h, Camera.h) and
 * INTEGRATION.md shows the binding a maintainer of the reference would add.
 *
 * Conventions
 *   - every function returns PT_OK (0) or a negative PT_E* code; pt_last_error() returns a
 *     thread-local message for the last failure on the calling thread.  The reference's
 *     gpuErrchk (include/CudaErrorCheck.h:6-14) prints "GPUassert: <msg> <file> <line>"
 *     and exit()s; the C++ look-alikes reproduce that on a non-zero return.
 *   - "d_" pointers are device (HBM) addresses of the currently selected device; they may
 *     be foreign allocations (e.g. a torch CUDA tensor's data_ptr, as main.cu:104,134 does).
 *   - output layout is the reference's: float32 [row][col][14], channels
 *     0-2 colour RGB, 3-5 normal XYZ, 6-8 albedo RGB, 9 depth, 10 colourVar, 11 normalVar,
 *     12 albedoVar, 13 depthVar (src/pathtrace.cu:240-254).
 *   - there is NO CPU fallback: without a usable HIP device every compute entry point
 *     fails with PT_ENODEVICE / PT_EHIP.
 */
#ifndef PTCORE_H
#define PTCORE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PT_ABI_VERSION 6

enum {
  PT_OK = 0,
  PT_EINVAL = -1,    /* bad argument                                         */
  PT_EHIP = -2,      /* a HIP runtime call failed (message has hipGetErrorString) */
  PT_ENODEVICE = -3, /* no HIP device visible                                */
  PT_ENOMEM = -4,    /* host allocation failed                               */
  PT_ELIMIT = -5,    /* scene does not fit the kernel's LDS staging budget   */
  PT_ECOMM = -6,     /* RCCL could not be loaded or a collective call failed (multi-GPU) */
  PT_ETIMEOUT = -7,  /* a multi-GPU frame did not complete in time; the exchange was aborted */
  PT_EKERNEL = -8    /* the kernel itself reported a failure (sample-chunk chain broken) for a frame that was  */
                     /*   ENQUEUED: that frame is incomplete -- the pixel blocks concerned were left untouched, */
                     /*   their generator state included (re-seed with pt_renderer_reset_rng / _set_rng_state   */
                     /*   to rejoin the reference's stream); sample chunking is off from then on.  A frame      */
                     /*   rendered with pt_renderer_render is completed by the call itself and returns PT_OK.   */
};

/* struct Sphere, include/Scene.h:7-14 -- same 40-byte layout, so a reference
 * `Sphere*` can be passed unchanged. */
typedef struct pt_sphere {
  float radius;
  float pos[3];
  float emission[3];
  float color[3];
} pt_sphere;

enum {
  PT_RNG_XORWOW = 0, /* cuRAND XORWOW, seed = pixel id (src/pathtrace.cu:265): reference parity */
  PT_RNG_PHILOX = 1  /* counter-based Philox4x32-10 keyed on (seed, frame); stateless           */
};

enum { PT_LAYOUT_INTERLEAVED = 0, PT_LAYOUT_PLANAR = 1 };

/* Options that the reference hard-codes or keeps in Renderer; zero-initialise then call
 * pt_renderer_opts_default(). */
typedef struct pt_renderer_opts {
  int32_t max_bounces;  /* MAX_BOUNCES, src/pathtrace.cu:7 (default 5)                     */
  int32_t rng_mode;     /* PT_RNG_*  (default PT_RNG_XORWOW)                               */
  uint64_t seed;        /* xorwow: added to the pixel id (0 = reference); philox: key      */
  int32_t row_begin;    /* rows [row_begin,row_end) of the width x height image are        */
  int32_t row_end;      /*   rendered by this renderer (multi-GPU row tiling); 0,0 = all   */
  int32_t persist_rng;  /* xorwow only: keep per-pixel generator state across Render()     */
                        /*   calls like Renderer::d_states (Renderer.h:17,37; pathtrace.cu:212,256); default 1 */
  int32_t variant;      /* kernel variant, all produce identical bits: -1 (default) = automatic    */
                        /*   (by scene size, tile size and generator), 0 = literal transcription,  */
                        /*   6, 8, 9, 10, 13, 14 see DESIGN.md section 3 (1-5, 7, 11, 12: libptcore_lab.so) */
  int32_t layout;       /* PT_LAYOUT_INTERLEAVED (default): the reference's [row][col][14] buffer  */
                        /*   (pathtrace.cu:240-254); PT_LAYOUT_PLANAR: [14][rows][width] of this   */
                        /*   renderer's tile -- same values, channel-first like a torch NCHW tensor */
  int32_t fast_math;    /* 0 (default): the bit-exact kernels.  1: the TOLERANCED fast mode -- same algorithm and */
                        /*   generator streams, FMA contraction allowed (nvcc's default for the reference), FP32-only   */
                        /*   cancellation-free intersectSphere, hardware rsq/sin/cos.  Distance from the reference's    */
                        /*   ORACLE at equal seeds, 256 x 256 (tests/test_fast_mode_gpu.py T5, profiles/r04/             */
                        /*   fast_vs_oracle.json): 1 spp -- albedo identical in >= 99.8 % of the pixels (measured:     */
                        /*   all), normals p99.9 6e-5 (L-inf 5e-4), depth p99.9 1e-5 relative; 64 spp -- colour beyond  */
                        /*   1e-4 in 1.6 % of the pixels (bound 3 %; median 0, L-inf 0.23: a ray that rounds onto       */
                        /*   another surface), normals 0.13 %, albedo 0.09 %; image means within 4 standard errors.      */
                        /*   NOT the north star's per-pixel L-inf 1e-4: only the exact kernels (bit-equal) meet that.   */
  int32_t chunks;       /* sample chunking (scheduling only, same bits): 0 (default) = automatic -- long frames of few   */
                        /*   workgroups split every pixel block's samples over several chained workgroups of one launch */
                        /*   (kernel map); 1 = never; 2..16 = that many chunks where the kernel supports it. */
                        /*   Costs scratch HBM (104 B per tile pixel) and hand-over traffic.  Env PT_CHUNKS overrides 0. */
  int32_t reserved;     /* must be 0 */
} pt_renderer_opts;

typedef struct pt_renderer pt_renderer; /* opaque; replaces class Renderer's private state, Renderer.h:10-20 */

/* ---- library / device -------------------------------------------------------------- */
int pt_abi_version(void);
/* 16 hex digits identifying the compiled sources + flags of this library (csrc/Makefile): measured
 * counters under profiles/ name the build they belong to. */
const char* pt_build_fingerprint(void);
const char* pt_last_error(void);
/* cudaSetDevice(cudaDevice), src/main.cu:86 */
int pt_set_device(int device);
int pt_device_count(int* count);
/* name + CU count of the selected device (for bench / logs) */
int pt_device_info(char* name, size_t name_len, int* compute_units, int* clock_khz);

/* ---- device memory: OutputBuffer / Scene allocations --------------------------------- */
/* cudaMalloc in OutputBuffer::AllocateGPU (OutputBuffer.h:73-74) and Scene() (Scene.h:36) */
int pt_malloc(void** d_ptr, size_t bytes);
/* cudaFree in OutputBuffer::FreeGPU (OutputBuffer.h:108-109) */
int pt_free(void* d_ptr);
/* cudaMemcpy H2D in Scene() (Scene.h:37) */
int pt_memcpy_h2d(void* d_dst, const void* h_src, size_t bytes);
/* cudaMemcpy D2H in OutputBuffer::CopyFromGPU (OutputBuffer.h:49-50) */
int pt_memcpy_d2h(void* h_dst, const void* d_src, size_t bytes);
int pt_memset(void* d_ptr, int value, size_t bytes);
int pt_device_synchronize(void);

/* ---- Renderer ------------------------------------------------------------------------ */
void pt_renderer_opts_default(pt_renderer_opts* opts);

/* Renderer::Renderer(width, height, samplesPerPixel, numThreads), Renderer.h:23-46.
 * Allocates the per-pixel generator state and runs setup_random (pathtrace.cu:259-266)
 * when rng_mode == PT_RNG_XORWOW && persist_rng.  threads_per_block is accepted for CLI
 * compatibility (main.cu:21,34) and ignored: the kernel picks its own workgroup shape.
 * opts may be NULL (= defaults). */
int pt_renderer_create(int width, int height, int samples_per_pixel, int threads_per_block,
                       const pt_renderer_opts* opts, pt_renderer** out);

/* Renderer::~Renderer(), Renderer.h:48-53 */
int pt_renderer_destroy(pt_renderer* r);

/* Renderer::Render(OutputBuffer d_buffer, const Scene& d_scene, const Camera& camera),
 * Renderer.h:55-76.  d_out points at the first float of row `row_begin` (for a full-frame
 * renderer: the buffer base).  basis = the 4 corner directions of
 * Camera::getEyeRayBasis (Camera.h:125-149), eye = camera.Position.  Synchronous like the
 * reference (returns after the stop event), *ms_out = kernel-only milliseconds from a
 * hipEvent pair (Renderer.h:63-75).  ms_out may be NULL.  (A frame whose sample-chunk chain broke -- never observed; see
 * PT_EKERNEL -- is repaired in place by a second launch: *ms_out then holds both launches INCLUDING the wait limit, default
 * 4 s, a line goes to stderr, pt_last_error() carries the same text and pt_renderer_check counts the frame.) */
int pt_renderer_render(pt_renderer* r, float* d_out, const pt_sphere* d_spheres, int n_spheres,
                       const float basis[12], const float eye[3], float* ms_out);

/* Same launch without host synchronisation, on the caller's HIP stream (hipStream_t passed
 * as void*; NULL = the default stream).  Used by bench.py and the multi-GPU driver so the
 * kernel overlaps with the caller's copies/collectives and can be timed with the caller's
 * events.  The camera (60 B) travels as kernel arguments: no H2D copy, no allocation.
 * One renderer's launches always execute in submission order: the renderer owns per-frame device
 * scratch (generator state, grid tables), so a launch on a different stream than the previous one
 * first waits (hipStreamWaitEvent) for that previous launch.  Frames that should overlap need
 * separate renderers. */
int pt_renderer_enqueue(pt_renderer* r, float* d_out, const pt_sphere* d_spheres, int n_spheres,
                        const float basis[12], const float eye[3], void* hip_stream);

/* n_frames frames with KNOWN cameras: what n_frames calls of pt_renderer_enqueue(r, d_out + f * out_stride_floats, ..., bases +
 * 12 f, eyes + 3 f, stream) do -- the body of the reference's frame loop (src/main.cu:146-177, `renderer.Render(...)` per
 * iteration) for a scripted fly-through or a pose sweep (collect_data.py) -- same frames, same persisted generator state
 * afterwards (src/pathtrace.cu:212,256), bit for bit.  For the reference's scene in the reference configuration (9 spheres, 5 or
 * 8 bounces, interleaved layout: the interactive shape, whose single frame is one round of waves -- all ramp and tail) the frames
 * go out as ONE launch per 32: a workgroup keeps its pixels for the whole batch and loops over the frames, the generator staying
 * in its registers from frame to frame (the counter-based one is re-keyed per frame) -- no state traffic, one ramp and one tail
 * per batch.  Elsewhere it IS the loop of single enqueues.
 * d_vertices != NULL: frame f also writes its display vertices to d_vertices + f * vtx_stride_floats (pt_renderer_set_display
 * for the batch).  Frames that would share a buffer (out_stride_floats below one tile, a set_display buffer without
 * d_vertices) are rendered one by one.  bases / eyes are host arrays, read before the call returns. */
int pt_renderer_enqueue_frames(pt_renderer* r, int n_frames, float* d_out, size_t out_stride_floats, float* d_vertices,
                               size_t vtx_stride_floats, const pt_sphere* d_spheres, int n_spheres, const float* bases,
                               const float* eyes, void* hip_stream);

/* Status of the frames enqueued so far (wait != 0: block until the last one's status word has arrived): PT_OK, or
 * PT_EKERNEL once for a frame whose sample-chunk chain broke (see PT_EKERNEL; the next pt_renderer_enqueue /
 * pt_renderer_render on the renderer would report it otherwise).  Call it after synchronising on the stream and
 * before pt_renderer_destroy, which reports nothing.  *repaired_frames (may be NULL) receives the number of frames
 * whose broken chain pt_renderer_render has repaired in place since the renderer was created (normally 0). */
int pt_renderer_check(pt_renderer* r, int wait, uint32_t* repaired_frames);

/* Renderer::Render + Denoiser::Denoise in ONE kernel (the body of the reference's interactive loop, src/main.cu:148,175):
 * from now on every frame of this renderer also writes the display vertices denoise_kernel (src/denoise.cu:9-29) derives
 * from it -- (col, width - row, RGBA8 {r,g,b,1} packed in a float) per pixel -- straight from the registers that hold the
 * pixel's colour: no second launch, no 12 B per pixel read back.  d_vertices: device float [rows of this renderer][width][3],
 * first vertex = first pixel of row_begin (like d_out); NULL switches it off.  Bit-identical to pt_display_pack(frame). */
int pt_renderer_set_display(pt_renderer* r, float* d_vertices);

/* Frame counter used by the philox key; incremented by every render/enqueue. */
int pt_renderer_set_frame(pt_renderer* r, uint32_t frame);
/* Re-run setup_random (pathtrace.cu:259-266): generator state as after construction. */
int pt_renderer_reset_rng(pt_renderer* r);
/* Copy the xorwow state ({d,v0..v4} = 6 x uint32 per tile pixel) to / from the host. */
int pt_renderer_get_rng_state(pt_renderer* r, uint32_t* h_state, size_t n_words);
int pt_renderer_set_rng_state(pt_renderer* r, const uint32_t* h_state, size_t n_words);
/* Static facts about the compiled kernel the renderer will launch. */
typedef struct pt_kernel_info {
  int32_t block_threads;
  int32_t grid_blocks;
  int32_t lds_bytes;
  int32_t num_vgprs;    /* from hipFuncGetAttributes */
  int32_t reserved0;    /* always 0 (was num_sgprs up to ABI 5: hipFuncGetAttributes does not report scalar registers) */
  int32_t scratch_bytes;
  int32_t max_spheres;  /* LDS staging limit for this variant (2^26 where larger scenes are not staged) */
  int32_t variant;      /* the variant the next launch will use (resolves the automatic choice) */
} pt_kernel_info;
int pt_renderer_kernel_info(pt_renderer* r, int n_spheres, pt_kernel_info* info);

/* ---- one frame over several GPUs of one node ------------------------------------------------- */
/* The reference selects ONE device per process (cudaSetDevice(cudaDevice), src/main.cu:86) and its
 * Renderer covers the whole image with one launch (Renderer.h:29-33,69).  pt_mgpu_* is the same
 * Renderer boundary over N devices: the image is cut into N contiguous row blocks (pixels are
 * independent and every generator is keyed on the global pixel id, pathtrace.cu:206,265, so the
 * result does not depend on the cut), one host thread per device renders its block, and ONE
 * exchange step per frame places the blocks in the caller's frame on devices[0]: grouped RCCL
 * ncclRecv x (N-1) on the root straight into the frame at the tile offsets | one ncclSend per peer.
 * Single process, no launcher; RCCL is dlopen'ed on first use. */
enum {
  PT_GATHER_AUTO = 0,      /* RCCL between distinct devices, peer copies if ranks share a device   */
  PT_GATHER_RCCL = 1,      /* ncclGroupStart{ncclRecv x (N-1) | ncclSend}ncclGroupEnd over xGMI     */
  PT_GATHER_PEER_COPY = 2  /* hipMemcpyPeerAsync of each tile on its own stream (SDMA over xGMI)    */
};
typedef struct pt_mgpu_opts {
  int32_t gather;          /* PT_GATHER_*                                                          */
  int32_t force_exchange;  /* 1: even the root's tile is rendered into a tile buffer and travels   */
                           /*    through the exchange step (self send/recv): exercises the whole   */
                           /*    multi-GPU path on a single-GPU machine.  Env PT_FORCE_MGPU=1.      */
  int32_t timeout_ms;      /* a frame not complete after this long fails with PT_ETIMEOUT and the  */
                           /*    communicator is aborted (0 = wait forever).  Env PT_MGPU_TIMEOUT_MS, default 60000 */
  int32_t bands;           /* row bands a rank's tile is rendered in: band b's transfer overlaps band b+1's   */
                           /*    kernel, only the last band's is exposed.  0 = automatic (bands of at least    */
                           /*    eight one-lane waves per SIMD, at most 8; 1 when no tile crosses a link), 1..64. */
                           /*    Env PT_MGPU_BANDS.  Same bits whatever the value.                               */
} pt_mgpu_opts;
typedef struct pt_mgpu pt_mgpu; /* opaque */

/* Defaults, with the environment overrides PT_FORCE_MGPU, PT_MGPU_TIMEOUT_MS, PT_MGPU_GATHER=rccl|copy, PT_MGPU_BANDS. */
void pt_mgpu_opts_default(pt_mgpu_opts* opts);
/* Renderer::Renderer over n_gpus devices (devices == NULL: 0..n_gpus-1; devices[0] is the root that
 * owns the caller's frame and scene).  opts as for pt_renderer_create, with row_begin = row_end = 0;
 * mopts may be NULL.  Creates per device: a renderer for its row block (generator state included),
 * a stream, a tile buffer; and one RCCL communicator over all of them (ncclCommInitAll). */
int pt_mgpu_create(int n_gpus, const int* devices, int width, int height, int samples_per_pixel,
                   int threads_per_block, const pt_renderer_opts* opts, const pt_mgpu_opts* mopts,
                   pt_mgpu** out);
int pt_mgpu_destroy(pt_mgpu* m);
/* Error behaviour of pt_mgpu_render: a rank that cannot render still posts its part of the exchange (nobody waits for
 * a tile that never comes) and the call reports the first failure.  After PT_ETIMEOUT, or any failure in which a rank lost
 * its communicator (ncclCommAbort), the object is dead: every later pt_mgpu_render returns PT_ECOMM and pt_mgpu_destroy is
 * the only call it still accepts.  Other failures (PT_EINVAL, PT_ELIMIT, PT_EHIP from a rank's render) leave it usable. */
/* Renderer::Render (Renderer.h:55-76) for the whole frame: d_out ([height][width][14]) and
 * d_spheres live on devices[0] and must be complete (the call does not order itself after the
 * caller's streams); the scene is replicated to the other devices by peer copies (360 B .. 40 KB).
 * Synchronous: returns when the frame is assembled.  *ms_out = end-to-end wall milliseconds
 * (render + exchange); per-tile kernel times: pt_mgpu_tile. */
int pt_mgpu_render(pt_mgpu* m, float* d_out, const pt_sphere* d_spheres, int n_spheres,
                   const float basis[12], const float eye[3], float* ms_out);
/* Row block, device and last kernel time of a rank (any out pointer may be NULL). */
int pt_mgpu_tile(pt_mgpu* m, int rank, int* device, int* row_begin, int* row_end, float* kernel_ms);
/* Timing of the last pt_mgpu_render (any out pointer may be NULL): bands per tile, the longest rank's render time (first
 * launch to last band done, hipEvent pair) and what the exchange added on top of it (end-to-end wall time minus that). */
int pt_mgpu_frame_stats(pt_mgpu* m, int* bands, float* render_ms, float* exposed_ms);
/* Name of the exchange backend in use (for logs / bench). */
int pt_mgpu_backend(pt_mgpu* m, char* name, size_t name_len);

/* ---- display packing ------------------------------------------------------------------- */
/* Denoiser::Denoise -> denoise_kernel (include/Denoiser.h:29-52, src/denoise.cu:9-29): despite the
 * name, the reference's "denoiser" only prepares the frame for its point-sprite display: colour
 * clamped to [0,1] and packed as RGBA8 {r,g,b,1} into ONE float, written with the pixel's screen
 * position as the vertex triple (col, width - row, packed) into d_vertices[row][col][3].  The
 * OpenGL buffer the reference maps (GLPixelBuffer) is replaced by a plain device pointer.
 * Asynchronous on hip_stream (NULL = default stream). */
int pt_display_pack(const float* d_buffer, int width, int height, float* d_vertices, void* hip_stream);

/* ---- host-side inputs of the path ------------------------------------------------------ */
/* The 9 spheres Scene() hard-codes, include/Scene.h:26-34 (host array). */
int pt_scene_cornell(pt_sphere out[9]);
/* Seeded random-sphere scene for BASELINE.json config 4 (no reference counterpart):
 * n spheres inside the Cornell box volume; with_walls != 0 appends nothing but makes the
 * first 6 entries the wall spheres of Scene.h:26-31 and the 7th the light (closed scene). */
int pt_scene_random(int n, uint64_t seed, int with_walls, pt_sphere* out);
/* Camera::updateCameraVectors + Camera::getEyeRayBasis, include/Camera.h:125-149,153-164,
 * restated without glm (float32, same operation order as glm 0.9.8). */
int pt_camera_basis(const float pos[3], float yaw_deg, float pitch_deg, int width, int height,
                    float basis_out[12]);
/* Same with an explicit WorldUp: the scalar constructor Camera(posX, posY, posZ, upX, upY, upZ, yaw, pitch),
 * include/Camera.h:63-70 (pt_camera_basis uses the (0, 1, 0) of the vector constructor, Camera.h:58). */
int pt_camera_basis_up(const float pos[3], float yaw_deg, float pitch_deg, const float world_up[3],
                       int width, int height, float basis_out[12]);

#ifdef __cplusplus
}
#endif
#endif /* PTCORE_H */
