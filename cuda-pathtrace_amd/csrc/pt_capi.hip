// pt_capi.hip -- the C ABI of include/ptcore.h on top of the HIP runtime.
// Host code only; device code is in pt_kernel.hip.  No CPU fallback exists: every compute
// entry point fails with PT_ENODEVICE / PT_EHIP when no gfx950 device is usable.
#include <hip/hip_runtime.h>
#include <math.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include "pt_internal.h"
#include "pt_kernel.h"

namespace {
thread_local char g_err[512] = "";
}

int pt_fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

#define PT_HIP(call)                                                                              \
  do {                                                                                            \
    hipError_t e_ = (call);                                                                       \
    if (e_ != hipSuccess)                                                                         \
      return pt_fail(e_ == hipErrorNoDevice ? PT_ENODEVICE : PT_EHIP, "%s: %s (%s:%d)", #call,    \
                     hipGetErrorString(e_), __FILE__, __LINE__);                                  \
  } while (0)

struct pt_renderer {
  int width, height, spp, threads_per_block;
  pt_renderer_opts opts;
  uint32_t tile_pixels;
  uint32_t frame;
  uint32_t* d_state;  // xorwow persistent state (Renderer::d_states), or nullptr
  hipEvent_t ev_start, ev_stop;
  int device;
  int launch_variant;      // what fill_args chose for the launch being prepared
  // automatic variant choice (opts.variant == PT_VARIANT_AUTO)
  bool auto_variant;
  uint64_t resident_pixels; // one-lane-per-pixel lanes the device holds at four waves per SIMD
  double waves_per_simd;    // one-lane-per-pixel waves of the tile per SIMD
  bool small_tile;         // fewer than three one-lane-per-pixel waves per SIMD (many-sphere scenes: four lanes per pixel pay there)
  bool spec_ok;            // variant 8's speculation has not been failing on this scene
  uint32_t* d_fail;        // device counter written by variant 8
  uint32_t* h_fail;        // pinned host copy, valid once ev_fail has completed
  hipEvent_t ev_fail;
  bool fail_pending;
  uint32_t* d_accel;       // variant 11's grid tables (rebuilt on the device before every frame)
  uint32_t* d_chunk;       // sample chunking (pt_kernel.hip): PT_CHUNK_WORDS words per tile pixel + one flag per pixel block; allocated
                           // by the first launch that chunks (never for renderers whose kernels do not), or null
  uint32_t chunks;         // how many chunks a frame of this renderer is split into when the kernel supports it (0 = off): variant 6 ...
  uint32_t chunks13;       // ... variant 13 ...
  uint32_t chunks_split;   // ... and the split kernels (variants 8, 9)
  uint64_t chunk_wait_ticks;  // how long a chunk waits for its predecessor (wall-clock ticks of the device)
  long chunk_wait_ms;         // ... the same in milliseconds (env PT_CHUNK_TIMEOUT_MS, default 4000)
  uint32_t repaired;          // frames whose broken chunk chain pt_renderer_render repaired (pt_renderer_check reports it)
  float* d_vertices;          // pt_renderer_set_display: the display vertices every frame also writes, or null
  uint32_t* d_err;         // device error word (PT_DEVERR_*), raised by a kernel that could not go on correctly
  uint32_t* h_err;         // pinned host copy, valid once ev_err has completed
  hipEvent_t ev_err;
  bool err_pending;        // a chunked launch's error word is on its way to h_err
  uint32_t debug;          // PT_DEBUG_* (lab library only, env PT_LAB_DEBUG)
  // The renderer owns single-instance device scratch (generator state, d_accel, d_fail): launches of one
  // renderer must execute in submission order even when the caller alternates streams.
  hipEvent_t ev_last;      // recorded after every launch on the stream it went to
  hipStream_t last_stream;
  bool have_last;
};

// Order this launch after the renderer's previous one if that went to a different stream.
static int order_after_last(pt_renderer* r, hipStream_t stream) {
  if (r->have_last && r->last_stream != stream) PT_HIP(hipStreamWaitEvent(stream, r->ev_last, 0));
  return PT_OK;
}

static int mark_last(pt_renderer* r, hipStream_t stream) {
  PT_HIP(hipEventRecord(r->ev_last, stream));
  r->last_stream = stream;
  r->have_last = true;
  return PT_OK;
}

// ---- which kernel for a small scene: a cost model instead of a table of bands (round 4) ---------------------------------
// Three kernels render the reference's scene bit-identically: variant 6 (one lane per pixel), 9 (two) and 8 (four).  More lanes
// per pixel = more, shorter waves: they fill a tile that gives each SIMD only a wave or two, and they pay for it with the
// generator skip-ahead and the record exchange.  Per kernel, three measured constants describe a SIMD that holds k of its
// waves (ms per 1024 samples of 5 bounces): L, what one wave needs alone (latency-bound), I, what each wave adds once the SIMD
// is issue-bound -- time(k) = (L^4 + (k I)^4)^(1/4), the smooth maximum of the two regimes -- and the round size, the waves a
// SIMD works on at a time when the frame is too short for sample chunking (then the tile is whole rounds plus a rest; a
// chunked frame fills its tail and behaves as one round of any size).  Unchunked waves run `unchunked` times slower (their
// tail is not filled) and every wave costs `per_wave` of prologue.  The tile has w one-lane waves per SIMD, hence
// ceil(lanes * w) waves of the kernel in question.  Fitted to profiles/r04/tile_policy_sweep.txt (row tiles of the headline
// frame, 1024 spp, both generators: the model's choice is within 0.5 % of the measured optimum at every point) and
// profiles/r04/short_frames_*.txt (256^2 ... 640^2 at 4 ... 256 spp: within 2.3 %); tests/test_policy_model.py keeps it so.
struct KernelCost {
  int variant, lanes;
  double L, I;
  int round;
  double unchunked, per_wave;
};
static const KernelCost kKernelCost[2][3] = {
    // cuRAND XORWOW (the split kernels re-generate their partners' draws)
    {{6, 1, 6.58, 2.97, 5, 1.031, 0.00666}, {8, 4, 1.85, 0.868, 5, 1.076, 0.00248}, {9, 2, 3.40, 1.60, 8, 1.108, 0.00448}},
    // Philox (counter-based: no skip-ahead)
    {{6, 1, 6.67, 3.05, 5, 1.031, 0.00666}, {8, 4, 1.75, 0.810, 8, 1.076, 0.00248}, {9, 2, 3.35, 1.60, 8, 1.108, 0.00448}}};
static const double kLaunchMs = 0.012;

static double predicted_ms(const KernelCost& c, double w, int spp, int bounces, bool chunked) {
  auto simd = [&](double k) { return c.I > 0 ? pow(pow(c.L, 4) + pow(k * c.I, 4), 0.25) : 0.0; };
  const double k = ceil(c.lanes * w - 1e-9), u = chunked ? 1.0 : c.unchunked;
  double t = simd(k);
  if (!chunked && k > c.round) {
    const double full = floor(k / c.round), rest = k - full * c.round;
    t = full * simd(c.round) + (rest > 0 ? simd(rest) : 0.0);
  }
  return kLaunchMs + k * c.per_wave + u * t * spp / 1024.0 * bounces / 5.0;
}

// the cheapest of variants 6, 8 (and 9 where it has a build: the reference configuration) for this tile
// chunked_mask: bit v set = a launch of variant v would chunk its samples (the regime is a property of the renderer --
// opts.chunks, PT_CHUNKS, a failed hand-over allocation, a broken chain all switch chunking off -- not of spp alone)
static int cheapest_variant(int rng_mode, double w, int spp, int bounces, bool with9, uint32_t chunked_mask) {
  int best = PT_DEFAULT_VARIANT;
  double best_ms = 1e300;
  for (const KernelCost& c : kKernelCost[rng_mode == PT_RNG_PHILOX ? 1 : 0]) {
    if (c.variant == 9 && !with9) continue;
    const double ms = predicted_ms(c, w, spp, bounces, ((chunked_mask >> c.variant) & 1u) != 0u);
    if (ms < best_ms) best = c.variant, best_ms = ms;  // (variant 6 comes first: it keeps a tie)
  }
  return best;
}

static int effective_variant(pt_renderer* r, int n_spheres) {
  if (r->opts.fast_math) return PT_VARIANT_FAST;
  if (!r->auto_variant) return r->opts.variant;
  if (r->fail_pending && hipEventQuery(r->ev_fail) == hipSuccess) {
    r->fail_pending = false;
    if (*r->h_fail > r->tile_pixels / 50u) r->spec_ok = false;  // > 2 % of the pixels left speculative mode: an open scene
  }
  // Many-sphere scenes: a bounce is n sphere tests; what matters is WHICH spheres are tested (the uniform grid with pooled
  // tests, variant 13: 160 ... 2048 spheres, from 72 on tiles that fill the chip) and that lanes whose path left the scene do not idle (path regeneration, variant
  // 10); a small closed tile still gains from four lanes per pixel until the speculation feedback says the scene is open.
  // (above ~1200 spheres the cell table wants more LDS than half a CU's: the same kernel with 1024-thread workgroups, one per CU --
  // 1024^2 x 32 spp closed / open, variant 13 | 14: 1200 spheres 13.8 / 4.3 | 12.3 / 3.6 ms, 1500: 18.0 / 6.3 | 15.6 / 5.0,
  // 2000: 69.4 / 34.5 | 18.4 / 6.2; at 1000 spheres and 256 spp 71.1 / 22.6 | 78.7 / 25.1: profiles/r05/large_scenes.txt)
  if (n_spheres > PT_GRID_WIDE_MIN_SPHERES && n_spheres <= PT_GRID_MAX_SPHERES) return 14;
  if (n_spheres >= PT_GRID_MIN_SPHERES && n_spheres <= PT_GRID_MAX_SPHERES) return 13;
  if (n_spheres >= PT_GRID_MIN_SPHERES_LARGE_TILE && n_spheres < PT_GRID_MIN_SPHERES && !r->small_tile) return 13;
  const bool xorwow = r->opts.rng_mode == PT_RNG_XORWOW;
  if (n_spheres > PT_SCREEN_MAX_SPHERES) return (xorwow && r->small_tile && r->spec_ok && r->spp >= 8) ? 8 : 10;
  // splitting a pixel's samples over lanes needs samples to split, and (xorwow) a speculation that holds
  if (r->spp < 4 || (xorwow && !r->spec_ok)) return PT_DEFAULT_VARIANT;
  const bool planar = r->opts.layout == PT_LAYOUT_PLANAR;
  const bool ref = pt_kernel_ref_bounces(n_spheres, r->opts.max_bounces, 9, planar) != 0;
  uint32_t chunked_mask = 0u;
  for (int v : {6, 8, 9}) {
    const uint32_t c = v == 6 ? r->chunks : r->chunks_split;
    if (c >= 2u && pt_kernel_chunked(v, n_spheres, r->opts.max_bounces, planar, r->spp, c)) chunked_mask |= 1u << v;
  }
  return cheapest_variant(r->opts.rng_mode, r->waves_per_simd, r->spp, r->opts.max_bounces > 0 ? r->opts.max_bounces : 1, ref, chunked_mask);
}

// opts.chunks, or env PT_CHUNKS when that is 0: 0 = automatic, 1 = never, 2..PT_CHUNKS_MAX = that many (anything else: automatic)
static int requested_chunks(int opt) {
  int want = opt;
  if (want == 0) {
    const char* env = getenv("PT_CHUNKS");
    if (env && *env) want = atoi(env);
  }
  return (want < 0 || want > PT_CHUNKS_MAX) ? 0 : want;
}

// A chunk may be at most max_samples long (the worst chained wait must stay far inside the wait limit): more chunks for very
// long frames, none at all if even PT_CHUNKS_MAX are not enough.  Returns the chunk count to use, 0 = no chunking.
static uint32_t fit_chunks(int want, int spp, int max_samples) {
  while (want >= 2 && want < PT_CHUNKS_MAX && (spp + want - 1) / want > max_samples) want *= 2;
  if (want < 2 || want > PT_CHUNKS_MAX || (spp + want - 1) / want > max_samples) return 0u;
  return (uint32_t)want;
}

extern "C" {

int pt_abi_version(void) { return PT_ABI_VERSION; }
#ifndef PT_BUILD_FINGERPRINT
#define PT_BUILD_FINGERPRINT "unknown"
#endif
const char* pt_build_fingerprint(void) { return PT_BUILD_FINGERPRINT; }
const char* pt_last_error(void) { return g_err; }

int pt_set_device(int device) {
  PT_HIP(hipSetDevice(device));
  return PT_OK;
}

int pt_device_count(int* count) {
  if (!count) return pt_fail(PT_EINVAL, "pt_device_count: count is NULL");
  *count = 0;
  PT_HIP(hipGetDeviceCount(count));
  return PT_OK;
}

int pt_device_info(char* name, size_t name_len, int* compute_units, int* clock_khz) {
  int dev = 0;
  PT_HIP(hipGetDevice(&dev));
  hipDeviceProp_t prop;
  PT_HIP(hipGetDeviceProperties(&prop, dev));
  if (name && name_len) snprintf(name, name_len, "%s (%s)", prop.name, prop.gcnArchName);
  if (compute_units) *compute_units = prop.multiProcessorCount;
  if (clock_khz) *clock_khz = prop.clockRate;
  return PT_OK;
}

int pt_malloc(void** d_ptr, size_t bytes) {
  if (!d_ptr) return pt_fail(PT_EINVAL, "pt_malloc: d_ptr is NULL");
  PT_HIP(hipMalloc(d_ptr, bytes));
  return PT_OK;
}

int pt_free(void* d_ptr) {
  PT_HIP(hipFree(d_ptr));
  return PT_OK;
}

int pt_memcpy_h2d(void* d_dst, const void* h_src, size_t bytes) {
  PT_HIP(hipMemcpy(d_dst, h_src, bytes, hipMemcpyHostToDevice));
  return PT_OK;
}

int pt_memcpy_d2h(void* h_dst, const void* d_src, size_t bytes) {
  PT_HIP(hipMemcpy(h_dst, d_src, bytes, hipMemcpyDeviceToHost));
  return PT_OK;
}

int pt_memset(void* d_ptr, int value, size_t bytes) {
  PT_HIP(hipMemset(d_ptr, value, bytes));
  return PT_OK;
}

int pt_device_synchronize(void) {
  PT_HIP(hipDeviceSynchronize());
  return PT_OK;
}

void pt_renderer_opts_default(pt_renderer_opts* o) {
  if (!o) return;
  memset(o, 0, sizeof(*o));
  o->max_bounces = 5;
  o->rng_mode = PT_RNG_XORWOW;
  o->persist_rng = 1;
  o->variant = PT_VARIANT_AUTO;
  o->layout = PT_LAYOUT_INTERLEAVED;
  o->fast_math = 0;
  o->chunks = 0;
}

static int setup_random(pt_renderer* r) {
  if (!r->d_state) return PT_OK;
  PT_HIP(pt_launch_setup_random(r->d_state, r->width, r->opts.row_begin, r->tile_pixels, r->opts.seed, nullptr));
  // The reference launches setup_random unchecked and unsynchronised (Renderer.h:38) because its
  // only stream orders the next launch after it.  pt_renderer_enqueue may run on ANY caller stream
  // (possibly non-blocking with respect to the default stream), so finish the 10 us kernel here.
  PT_HIP(hipStreamSynchronize(nullptr));
  return PT_OK;
}

int pt_renderer_create(int width, int height, int samples_per_pixel, int threads_per_block,
                       const pt_renderer_opts* opts, pt_renderer** out) {
  if (!out) return pt_fail(PT_EINVAL, "pt_renderer_create: out is NULL");
  *out = nullptr;
  pt_renderer_opts o;
  if (opts) o = *opts; else pt_renderer_opts_default(&o);
  if (o.row_begin == 0 && o.row_end == 0) o.row_end = height;
  if (width <= 0 || height <= 0 || samples_per_pixel <= 0)
    return pt_fail(PT_EINVAL, "pt_renderer_create: width/height/samples must be positive (%d x %d x %d)", width, height,
                   samples_per_pixel);
  if (o.row_begin < 0 || o.row_end > height || o.row_begin > o.row_end)
    return pt_fail(PT_EINVAL, "pt_renderer_create: bad row range [%d,%d) for height %d", o.row_begin, o.row_end, height);
  if (o.max_bounces < 0 || o.max_bounces > 64) return pt_fail(PT_EINVAL, "pt_renderer_create: max_bounces %d", o.max_bounces);
  if (o.rng_mode != PT_RNG_XORWOW && o.rng_mode != PT_RNG_PHILOX)
    return pt_fail(PT_EINVAL, "pt_renderer_create: rng_mode %d", o.rng_mode);
  if (o.layout != PT_LAYOUT_INTERLEAVED && o.layout != PT_LAYOUT_PLANAR) return pt_fail(PT_EINVAL, "pt_renderer_create: layout %d", o.layout);
  if (o.fast_math != 0 && o.fast_math != 1) return pt_fail(PT_EINVAL, "pt_renderer_create: fast_math %d", o.fast_math);
  if (o.fast_math && o.variant != PT_VARIANT_AUTO)
    return pt_fail(PT_EINVAL, "pt_renderer_create: fast_math has one kernel; leave variant at -1");
  if (o.chunks < 0 || o.chunks > PT_CHUNKS_MAX || o.reserved != 0)
    return pt_fail(PT_EINVAL, "pt_renderer_create: chunks %d (0 = automatic, 1 = off, 2..%d), reserved %d (must be 0)", o.chunks, PT_CHUNKS_MAX, o.reserved);
  if (o.variant != PT_VARIANT_AUTO && !pt_kernel_has_variant(o.variant))
    return pt_fail(PT_EINVAL, "pt_renderer_create: kernel variant %d is not in this build (product variants: 0, 6, 8, 9, 10, 13; "
                              "the experiments 1-5, 7, 11, 12 live in libptcore_lab.so)", o.variant);
  // 32-bit pixel ids like the reference (pathtrace.cu:206): width*height must fit uint32
  if ((uint64_t)width * (uint64_t)height > 0xFFFFFFFFull) return pt_fail(PT_EINVAL, "pt_renderer_create: image too large");

  int ndev = 0;
  PT_HIP(hipGetDeviceCount(&ndev));
  if (ndev <= 0) return pt_fail(PT_ENODEVICE, "pt_renderer_create: no HIP device");

  pt_renderer* r = new (std::nothrow) pt_renderer();
  if (!r) return pt_fail(PT_ENOMEM, "pt_renderer_create: out of host memory");
  r->width = width;
  r->height = height;
  r->spp = samples_per_pixel;
  r->threads_per_block = threads_per_block;
  r->opts = o;
  r->tile_pixels = (uint32_t)(o.row_end - o.row_begin) * (uint32_t)width;
  r->frame = 0;
  r->d_state = nullptr;
  r->ev_start = r->ev_stop = nullptr;
  r->auto_variant = (o.variant == PT_VARIANT_AUTO);
  r->small_tile = false;
  r->resident_pixels = 0;
  r->waves_per_simd = 1e9;
  r->spec_ok = true;
  r->d_fail = nullptr;
  r->h_fail = nullptr;
  r->ev_fail = nullptr;
  r->fail_pending = false;
  r->d_accel = nullptr;
  r->d_chunk = nullptr;
  r->chunks = 0;
  r->chunks13 = 0;
  r->chunks_split = 0;
  r->chunk_wait_ticks = 0;
  r->chunk_wait_ms = 0;
  r->repaired = 0;
  r->d_vertices = nullptr;
  r->d_err = nullptr;
  r->h_err = nullptr;
  r->ev_err = nullptr;
  r->err_pending = false;
  r->debug = 0;
  r->ev_last = nullptr;
  r->last_stream = nullptr;
  r->have_last = false;
  hipError_t e = hipGetDevice(&r->device);
  if (e == hipSuccess) {
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, r->device);
    if (e == hipSuccess) {
      const uint64_t simds = (uint64_t)prop.multiProcessorCount * 4u;
      r->resident_pixels = simds * 4u * 64u;
      r->waves_per_simd = (double)r->tile_pixels / (double)(simds * 64u);
      r->small_tile = (uint64_t)r->tile_pixels < simds * 64u * 3u;
    }
  }
  if (e == hipSuccess) e = hipMalloc((void**)&r->d_fail, sizeof(uint32_t));
  if (e == hipSuccess) e = hipMemset(r->d_fail, 0, sizeof(uint32_t));
  if (e == hipSuccess) e = hipHostMalloc((void**)&r->h_fail, sizeof(uint32_t), hipHostMallocDefault);
  if (e == hipSuccess) { *r->h_fail = 0; e = hipEventCreateWithFlags(&r->ev_fail, hipEventDisableTiming); }
  if (e == hipSuccess) e = hipMalloc((void**)&r->d_accel, pt_kernel_accel_bytes());
  // Sample chunking pays on frames that make few rounds of long workgroups (tools/shape_sweep.py): at least 512 samples per
  // pixel and at most 8 one-lane waves per SIMD slot-round.  opts.chunks / env PT_CHUNKS: 0 = this policy, 1 = never, n = n chunks.
  // The hand-over buffer (104 B per tile pixel) is allocated by the first launch that really chunks (chunk_buffer()).
  if (e == hipSuccess) {
    // opts.chunks / env PT_CHUNKS, one reading for all three kernel families: 0 = the policies below, 1 = never, n = n chunks
    const int asked = requested_chunks(o.chunks);
    hipDeviceProp_t prop;
    const bool have_prop = hipGetDeviceProperties(&prop, r->device) == hipSuccess;
    int want = asked;
    if (want == 0 && r->spp >= 512 && r->tile_pixels > 0 && have_prop) {
      const uint64_t slots = (uint64_t)prop.multiProcessorCount * 4u * 4u * 64u;  // pixels resident at 4 waves per SIMD
      // The whole frame: two chunks (one hand-over per pixel; 49.95 -> 49.48 ms, eight: 49.40).  Row tiles -- what a rank of a multi-GPU
      // run renders -- are one or two rounds of waves and gain more from more chunks (tools/chunk_tile.py, profiles/r03/README.md;
      // chunks 1 / 2 / 4 / 6 / 8): 1/4 frame 13.03 / 13.31 / 12.71 / 13.03 / 12.62 ms, 1/2 frame 25.50 / 25.28 / 25.39 / 25.58 / 24.89,
      // 3/16 frame 10.31 / 10.05 / 10.15 / 9.81 / 10.61 (the steps: chunks x waves per SIMD against the five resident).
      if ((uint64_t)r->tile_pixels <= 8u * slots)
        want = r->waves_per_simd > 8.5 ? PT_CHUNKS : (r->waves_per_simd > 3.25 ? PT_CHUNKS_TILE : PT_CHUNKS_SMALL_TILE);
    }
    r->chunks = fit_chunks(want, r->spp, PT_CHUNK_MAX_SAMPLES);
    // The pooled grid kernel (variant 13): 512-pixel workgroups, two resident per CU, 25 ms each at 1000 spheres x 256 spp --
    // a 1024^2 frame is FOUR rounds of them and its time is whatever the last round's stragglers make it (measured: 98 or 107 ms
    // from one frame to the next, profiles/r03/README.md).  Enough chunks for about thirty-two rounds, each at least 32 samples
    // long (closed / open at 256 spp, chunks 1: 98-109 / 38.4 ms, 2: 101.5 / 36.4, 4: 97.3 / 35.4, 8: 96.0 / 35.1, 16: 97.9 / 36.3).
    // Its samples are a hundred times as long as the nine-sphere kernel's (about 0.1 ms per sample and workgroup at 1000
    // spheres), so a chunk is capped at PT_CHUNK_MAX_SAMPLES_GRID samples: the worst chained wait -- all chunks of a block
    // co-resident, chunk k waits k chunk durations -- stays a fraction of a second.
    {
      int want13 = asked;
      if (want13 == 0 && r->tile_pixels > 0 && have_prop) {
        const uint64_t resident = (uint64_t)prop.multiProcessorCount * 2u;  // workgroups
        const uint64_t groups = ((uint64_t)r->tile_pixels + 511u) / 512u;
        const uint64_t rounds = (groups + resident - 1) / resident;
        want13 = rounds >= 32 ? 1 : (int)((32 + rounds - 1) / rounds);
        if (want13 > PT_CHUNKS_GRID_MAX) want13 = PT_CHUNKS_GRID_MAX;
        while (want13 >= 2 && r->spp / want13 < 32) want13 /= 2;
        // Round 5: a tile whose workgroups fill at most HALF the chip's slots gains nothing from chunk workgroups -- they are resident
        // at once and only wait for their predecessors, in slots the working ones could have had to themselves (rows 512..640 of
        // the 1000-sphere frame, closed / open: 8 chunks 29.1 / 19.1 ms, none 17.9 / 10.4; profiles/r05/cfg4_tile_chunks.txt)
        if (groups * 2u <= resident) want13 = 1;
      }
      r->chunks13 = fit_chunks(want13, r->spp, PT_CHUNK_MAX_SAMPLES_GRID);
    }
    // the split kernels (small tiles of the reference configuration: one or two rounds of waves by construction): four chunks
    // (tools/chunk_tile.py, profiles/r03/tile_policy_bitops.txt)
    r->chunks_split = fit_chunks(asked == 0 && r->spp >= 512 ? PT_CHUNKS_SPLIT : asked, r->spp, PT_CHUNK_MAX_SAMPLES);
    int khz = 0;  // s_memrealtime ticks per millisecond
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, r->device) != hipSuccess || khz <= 0) khz = 100000;
    long wait_ms = 4000;
    if (const char* env = getenv("PT_CHUNK_TIMEOUT_MS")) { if (*env) wait_ms = atol(env); }
    if (wait_ms < 1) wait_ms = 1;
    r->chunk_wait_ticks = (uint64_t)khz * (uint64_t)wait_ms;
    r->chunk_wait_ms = wait_ms;
#if PT_BUILD_EXPERIMENTS
    if (const char* env = getenv("PT_LAB_DEBUG")) r->debug = (uint32_t)strtoul(env, nullptr, 0);
#endif
  }
  if (e == hipSuccess) e = hipMalloc((void**)&r->d_err, sizeof(uint32_t));
  if (e == hipSuccess) e = hipMemset(r->d_err, 0, sizeof(uint32_t));
  if (e == hipSuccess) e = hipHostMalloc((void**)&r->h_err, sizeof(uint32_t), hipHostMallocDefault);
  if (e == hipSuccess) { *r->h_err = 0; e = hipEventCreateWithFlags(&r->ev_err, hipEventDisableTiming); }
  if (e == hipSuccess) e = hipEventCreateWithFlags(&r->ev_last, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreate(&r->ev_start);
  if (e == hipSuccess) e = hipEventCreate(&r->ev_stop);
  if (e == hipSuccess && o.rng_mode == PT_RNG_XORWOW && o.persist_rng && r->tile_pixels)
    e = hipMalloc((void**)&r->d_state, (size_t)r->tile_pixels * 6 * sizeof(uint32_t));  // Renderer.h:37
  if (e != hipSuccess) {
    int rc = pt_fail(PT_EHIP, "pt_renderer_create: %s", hipGetErrorString(e));
    pt_renderer_destroy(r);
    return rc;
  }
  int rc = setup_random(r);  // Renderer.h:38
  if (rc != PT_OK) {
    pt_renderer_destroy(r);
    return rc;
  }
  *out = r;
  return PT_OK;
}

int pt_renderer_destroy(pt_renderer* r) {
  if (!r) return PT_OK;
  if (r->d_state) (void)hipFree(r->d_state);  // Renderer.h:50
  if (r->d_accel) (void)hipFree(r->d_accel);
  if (r->d_chunk) (void)hipFree(r->d_chunk);
  if (r->d_err) (void)hipFree(r->d_err);
  if (r->h_err) (void)hipHostFree(r->h_err);
  if (r->ev_err) (void)hipEventDestroy(r->ev_err);
  if (r->d_fail) (void)hipFree(r->d_fail);
  if (r->h_fail) (void)hipHostFree(r->h_fail);
  if (r->ev_fail) (void)hipEventDestroy(r->ev_fail);
  if (r->ev_last) (void)hipEventDestroy(r->ev_last);
  if (r->ev_start) (void)hipEventDestroy(r->ev_start);
  if (r->ev_stop) (void)hipEventDestroy(r->ev_stop);
  delete r;
  return PT_OK;
}

// Does a launch of this renderer with this variant and scene chain a pixel's samples through several workgroups?  Allocates the
// hand-over buffer the first time the answer is yes; an allocation failure turns chunking off for good (it is a scheduling
// aid, never a reason for a renderer not to work).
static uint32_t chunks_of(const pt_renderer* r, int variant) {
  return (variant == 13 || variant == 14) ? r->chunks13 : (variant == 8 || variant == 9) ? r->chunks_split : r->chunks;
}

static bool chunk_buffer(pt_renderer* r, int variant, int n_spheres) {
  const uint32_t chunks = chunks_of(r, variant);
  if (chunks < 2u || !pt_kernel_chunked(variant, n_spheres, r->opts.max_bounces, r->opts.layout == PT_LAYOUT_PLANAR, r->spp, chunks))
    return false;
  if (r->d_chunk) return true;
  const size_t blocks = ((size_t)r->tile_pixels * 4 + PT_BLOCK_THREADS - 1) / PT_BLOCK_THREADS;  // (the four-lane kernel has the most pixel blocks)
  if (hipMalloc((void**)&r->d_chunk, ((size_t)PT_CHUNK_WORDS * r->tile_pixels + blocks) * sizeof(uint32_t)) != hipSuccess) {
    (void)hipGetLastError();  // not sticky: the launch that follows must not inherit it
    r->d_chunk = nullptr;
    r->chunks = 0;
    r->chunks13 = 0;
    r->chunks_split = 0;
    return false;
  }
  return true;
}

// The error word of an earlier chunked launch, once it has arrived (wait: block until it has).  *err receives the word (0 = the
// chain held) and is cleared on the device; a broken chain also switches chunking off for this renderer for good -- whatever
// broke it may do so again.
static int take_device_error(pt_renderer* r, bool wait, uint32_t* err) {
  *err = 0u;
  if (!r->err_pending) return PT_OK;
  if (wait) PT_HIP(hipEventSynchronize(r->ev_err));
  else if (hipEventQuery(r->ev_err) != hipSuccess) { (void)hipGetLastError(); return PT_OK; }
  r->err_pending = false;
  *err = *r->h_err;
  if (*err == 0u) return PT_OK;
  *r->h_err = 0u;
  PT_HIP(hipMemset(r->d_err, 0, sizeof(uint32_t)));
  r->chunks = 0;
  r->chunks13 = 0;
  r->chunks_split = 0;
  return PT_OK;
}

// ... as a status: PT_EKERNEL for a frame that was enqueued and is incomplete (pt_renderer_render repairs its own instead)
static int check_device_error(pt_renderer* r, bool wait) {
  uint32_t err = 0u;
  const int rc = take_device_error(r, wait, &err);
  if (rc != PT_OK || err == 0u) return rc;
  return pt_fail(PT_EKERNEL, "render: sample-chunk chain broken (device error word 0x%x): a workgroup waited %ld ms for its "
                             "predecessor in vain; that frame is incomplete (the pixel blocks concerned were left untouched, their "
                             "generator state included), chunking is now off for this renderer", err, r->chunk_wait_ms);
}

static int fill_args(pt_renderer* r, float* d_out, const pt_sphere* d_spheres, int n_spheres, const float basis[12],
                     const float eye[3], PixelKernelArgs* a) {
  if (!r) return pt_fail(PT_EINVAL, "render: renderer is NULL");
  if (!d_out && r->tile_pixels) return pt_fail(PT_EINVAL, "render: d_out is NULL");
  if (n_spheres < 0 || (n_spheres > 0 && !d_spheres)) return pt_fail(PT_EINVAL, "render: bad scene (%d spheres)", n_spheres);
  if (!basis || !eye) return pt_fail(PT_EINVAL, "render: basis/eye is NULL");
  const int variant = effective_variant(r, n_spheres);
  if (variant != PT_VARIANT_FAST && n_spheres > pt_kernel_max_spheres(variant))
    return pt_fail(PT_ELIMIT, "render: %d spheres exceed the LDS staging limit of %d", n_spheres,
                   pt_kernel_max_spheres(variant));
  a->fail_count = r->d_fail;
  a->accel = r->d_accel;
  a->scene_lds_f4 = 0;
  a->planar = r->opts.layout == PT_LAYOUT_PLANAR ? 1u : 0u;
  r->launch_variant = variant;
  a->out = d_out;
  a->spheres = d_spheres;
  a->rng_state = r->d_state;
  memcpy(a->basis, basis, sizeof(a->basis));
  memcpy(a->eye, eye, sizeof(a->eye));
  a->n_spheres = n_spheres;
  a->width = r->width;
  a->height = r->height;
  a->row_begin = r->opts.row_begin;
  a->tile_pixels = r->tile_pixels;
  a->spp = r->spp;
  a->max_bounces = r->opts.max_bounces;
  a->frame = r->frame;
  a->seed = r->opts.seed;
  // sample chunking: only for the kernels that chunk (reference configuration, variant 6), and only with a hand-over buffer
  const bool chunking = variant != PT_VARIANT_FAST && chunk_buffer(r, variant, n_spheres);
  a->chunks = chunking ? chunks_of(r, variant) : 0u;
  a->chunk_state = chunking ? r->d_chunk : nullptr;
  a->chunk_flag = chunking ? r->d_chunk + (size_t)PT_CHUNK_WORDS * r->tile_pixels : nullptr;
  // a frame that fits the chip in at most two rounds of workgroups is mostly tail: its waves set their priority by progress
  a->prio = (r->spp >= 4 && (uint64_t)r->tile_pixels <= 2u * r->resident_pixels) ? 1u : 0u;
  a->err_word = r->d_err;
  a->chunk_wait_ticks = r->chunk_wait_ticks;
  a->debug = r->debug;
  a->repair = 0u;
  a->vertices = r->d_vertices;
  return PT_OK;
}

// With the automatic policy a speculative launch is bracketed by the reset and the asynchronous
// read-back of its failure counter (consulted by a later effective_variant()).  The bracket sits
// OUTSIDE the event pair of pt_renderer_render so the returned kernel time is not distorted.
static hipError_t launch(pt_renderer* r, const PixelKernelArgs& a, hipStream_t stream) {
  if (r->launch_variant == PT_VARIANT_FAST) return pt_launch_fast_kernel(a, r->opts.rng_mode, stream);
  return pt_launch_pixel_kernel(a, r->opts.rng_mode, r->launch_variant, stream);
}

static int watch_begin(pt_renderer* r, hipStream_t stream, bool* watching) {
  *watching = r->auto_variant && (r->launch_variant == 8 || r->launch_variant == 9) && !r->fail_pending;
  if (*watching) PT_HIP(hipMemsetAsync(r->d_fail, 0, sizeof(uint32_t), stream));
  return PT_OK;
}

static int watch_error(pt_renderer* r, const PixelKernelArgs& a, hipStream_t stream) {
  if (a.chunks < 2u) return PT_OK;  // only chunked launches can raise the word
  PT_HIP(hipMemcpyAsync(r->h_err, r->d_err, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
  PT_HIP(hipEventRecord(r->ev_err, stream));
  r->err_pending = true;
  return PT_OK;
}

static int watch_end(pt_renderer* r, hipStream_t stream, bool watching) {
  if (!watching) return PT_OK;
  PT_HIP(hipMemcpyAsync(r->h_fail, r->d_fail, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
  PT_HIP(hipEventRecord(r->ev_fail, stream));
  r->fail_pending = true;
  return PT_OK;
}

int pt_renderer_enqueue(pt_renderer* r, float* d_out, const pt_sphere* d_spheres, int n_spheres, const float basis[12],
                        const float eye[3], void* hip_stream) {
  PixelKernelArgs a;
  if (!r) return pt_fail(PT_EINVAL, "render: renderer is NULL");
  int rc = check_device_error(r, false);  // an earlier asynchronous frame whose chain broke is reported here
  if (rc != PT_OK) return rc;
  rc = fill_args(r, d_out, d_spheres, n_spheres, basis, eye, &a);
  if (rc != PT_OK) return rc;
  if (r->tile_pixels == 0) return PT_OK;
  rc = order_after_last(r, (hipStream_t)hip_stream);
  if (rc != PT_OK) return rc;
  bool watching = false;
  rc = watch_begin(r, (hipStream_t)hip_stream, &watching);
  if (rc != PT_OK) return rc;
  PT_HIP(launch(r, a, (hipStream_t)hip_stream));
  rc = watch_end(r, (hipStream_t)hip_stream, watching);
  if (rc != PT_OK) return rc;
  rc = watch_error(r, a, (hipStream_t)hip_stream);
  if (rc != PT_OK) return rc;
  rc = mark_last(r, (hipStream_t)hip_stream);
  if (rc != PT_OK) return rc;
  r->frame++;
  return PT_OK;
}

// A batch of frames with known cameras (include/ptcore.h).  Where a frames kernel exists -- the reference's scene in the
// reference configuration, i.e. the interactive shape -- the batch goes out as launches of up to PT_FRAMES_MAX frames each;
// everywhere else, and whenever two frames would share a buffer, it is the loop of single-frame enqueues it stands for.
int pt_renderer_enqueue_frames(pt_renderer* r, int n_frames, float* d_out, size_t out_stride_floats, float* d_vertices,
                               size_t vtx_stride_floats, const pt_sphere* d_spheres, int n_spheres, const float* bases,
                               const float* eyes, void* hip_stream) {
  if (!r) return pt_fail(PT_EINVAL, "pt_renderer_enqueue_frames: renderer is NULL");
  if (n_frames < 0 || (n_frames > 0 && (!bases || !eyes))) return pt_fail(PT_EINVAL, "pt_renderer_enqueue_frames: bad arguments");
  if (n_frames == 0) return PT_OK;
  float* const saved_vertices = r->d_vertices;
  const size_t tile_floats = (size_t)r->tile_pixels * 14u, vtx_floats = (size_t)r->tile_pixels * 3u;
  const int variant = effective_variant(r, n_spheres);
  const bool batched = n_frames >= 2 && r->tile_pixels > 0 && variant != PT_VARIANT_FAST &&
                       pt_kernel_has_frames(variant, n_spheres, r->opts.max_bounces, r->opts.layout == PT_LAYOUT_PLANAR) &&
                       out_stride_floats >= tile_floats && (d_vertices ? vtx_stride_floats >= vtx_floats : saved_vertices == nullptr);
  int rc = PT_OK;
  for (int f0 = 0; f0 < n_frames && rc == PT_OK;) {
    const int m = batched ? (n_frames - f0 < PT_FRAMES_MAX ? n_frames - f0 : PT_FRAMES_MAX) : 1;
    float* out_f = d_out ? d_out + (size_t)f0 * out_stride_floats : nullptr;
    if (d_vertices) r->d_vertices = d_vertices + (size_t)f0 * vtx_stride_floats;
    if (m < 2) {
      rc = pt_renderer_enqueue(r, out_f, d_spheres, n_spheres, bases + 12 * (size_t)f0, eyes + 3 * (size_t)f0, hip_stream);
      f0 += 1;
      continue;
    }
    FramesKernelArgs fa;
    rc = check_device_error(r, false);
    if (rc != PT_OK) break;
    rc = fill_args(r, out_f, d_spheres, n_spheres, bases + 12 * (size_t)f0, eyes + 3 * (size_t)f0, &fa.base);
    if (rc != PT_OK) break;
    fa.base.chunks = 0u;  // (a batch's workgroups keep their pixel block: nothing is handed from workgroup to workgroup)
    fa.base.chunk_state = nullptr;
    fa.base.chunk_flag = nullptr;
    fa.frames = (uint32_t)m;
    fa.out_stride = out_stride_floats;
    fa.vtx_stride = vtx_stride_floats;
    for (int k = 0; k < m; k++) {
      memcpy(fa.cams[k], bases + 12 * (size_t)(f0 + k), 12 * sizeof(float));
      memcpy(fa.cams[k] + 12, eyes + 3 * (size_t)(f0 + k), 3 * sizeof(float));
    }
    rc = order_after_last(r, (hipStream_t)hip_stream);
    if (rc != PT_OK) break;
    {
      const hipError_t e = pt_launch_frames_kernel(fa, r->opts.rng_mode, (hipStream_t)hip_stream);
      if (e != hipSuccess) { rc = pt_fail(PT_EHIP, "pt_renderer_enqueue_frames: %s", hipGetErrorString(e)); break; }
    }
    rc = mark_last(r, (hipStream_t)hip_stream);
    r->frame += (uint32_t)m;
    f0 += m;
  }
  r->d_vertices = saved_vertices;
  return rc;
}

int pt_renderer_render(pt_renderer* r, float* d_out, const pt_sphere* d_spheres, int n_spheres, const float basis[12],
                       const float eye[3], float* ms_out) {
  PixelKernelArgs a;
  if (!r) return pt_fail(PT_EINVAL, "render: renderer is NULL");
  int rc = check_device_error(r, true);
  if (rc != PT_OK) return rc;
  rc = fill_args(r, d_out, d_spheres, n_spheres, basis, eye, &a);
  if (rc != PT_OK) return rc;
  if (ms_out) *ms_out = 0.0f;
  if (r->tile_pixels == 0) return PT_OK;
  rc = order_after_last(r, nullptr);
  if (rc != PT_OK) return rc;
  bool watching = false;
  rc = watch_begin(r, nullptr, &watching);
  if (rc != PT_OK) return rc;
  PT_HIP(hipEventRecord(r->ev_start, nullptr));  // Renderer.h:68
  PT_HIP(launch(r, a, nullptr));
  PT_HIP(hipEventRecord(r->ev_stop, nullptr));   // Renderer.h:70
  rc = watch_end(r, nullptr, watching);
  if (rc != PT_OK) return rc;
  rc = watch_error(r, a, nullptr);
  if (rc != PT_OK) return rc;
  PT_HIP(hipEventSynchronize(r->ev_stop));       // Renderer.h:72
  r->have_last = false;  // the launch has completed: nothing left to order against
  r->frame++;
  float ms = 0.0f;
  PT_HIP(hipEventElapsedTime(&ms, r->ev_start, r->ev_stop));
  // A synchronous frame whose chunk chain broke is completed HERE: the chunked launch left every pixel block either complete
  // or untouched, generator state included (pt_kernel.hip, chunk_wait), so one unchunked launch over the untouched blocks
  // makes the frame -- and the state the next frame starts from -- exactly what an unbroken launch would have left.
  uint32_t err = 0u;
  rc = take_device_error(r, true, &err);
  if (rc != PT_OK) return rc;
  if (err != 0u) {
    a.repair = a.chunks;  // the count a complete block's flag holds
    a.chunks = 0u;
    PT_HIP(hipEventRecord(r->ev_start, nullptr));
    PT_HIP(launch(r, a, nullptr));
    PT_HIP(hipEventRecord(r->ev_stop, nullptr));
    PT_HIP(hipEventSynchronize(r->ev_stop));
    float ms2 = 0.0f;
    PT_HIP(hipEventElapsedTime(&ms2, r->ev_start, r->ev_stop));
    ms += ms2;
    r->repaired++;
    // not an error (the frame is complete), but never silent either: the caller's frame time holds the whole wait
    snprintf(g_err, sizeof(g_err), "render: sample-chunk chain broken (device error word 0x%x) -- frame repaired in place by a second, unchunked "
             "launch over the untouched pixel blocks; the returned time includes the %ld ms wait; chunking is now off for this renderer", err, r->chunk_wait_ms);
    fprintf(stderr, "ptcore: %s\n", g_err);
  }
  if (ms_out) *ms_out = ms;
  return PT_OK;
}

int pt_renderer_check(pt_renderer* r, int wait, uint32_t* repaired_frames) {
  if (!r) return pt_fail(PT_EINVAL, "pt_renderer_check: renderer is NULL");
  if (repaired_frames) *repaired_frames = r->repaired;
  return check_device_error(r, wait != 0);
}

#if PT_BUILD_EXPERIMENTS
// lab library: the variant the C++ policy itself picks for a tile of the reference's scene (tests/test_policy_model.py compares
// it with the Python restatement of the argmin)
int pt_debug_policy_choice(int rng_mode, double waves_per_simd, int spp, int bounces, int with9, int chunked, int* variant) {
  if (!variant) return pt_fail(PT_EINVAL, "pt_debug_policy_choice: variant is NULL");
  *variant = cheapest_variant(rng_mode, waves_per_simd, spp, bounces, with9 != 0, chunked ? ((1u << 6) | (1u << 8) | (1u << 9)) : 0u);
  return PT_OK;
}
// lab library: the policy's cost model, for tests/test_policy_model.py (include/ptcore_lab.h)
int pt_debug_policy_ms(int rng_mode, int variant, double waves_per_simd, int spp, int bounces, double* ms) {
  if (!ms) return pt_fail(PT_EINVAL, "pt_debug_policy_ms: ms is NULL");
  for (const KernelCost& c : kKernelCost[rng_mode == PT_RNG_PHILOX ? 1 : 0])
    if (c.variant == variant) {
      *ms = predicted_ms(c, waves_per_simd, spp, bounces, spp >= 512);
      return PT_OK;
    }
  return pt_fail(PT_EINVAL, "pt_debug_policy_ms: no cost record for variant %d", variant);
}
#endif

int pt_renderer_set_display(pt_renderer* r, float* d_vertices) {
  if (!r) return pt_fail(PT_EINVAL, "pt_renderer_set_display: renderer is NULL");
  r->d_vertices = d_vertices;
  return PT_OK;
}

int pt_renderer_set_frame(pt_renderer* r, uint32_t frame) {
  if (!r) return pt_fail(PT_EINVAL, "pt_renderer_set_frame: renderer is NULL");
  r->frame = frame;
  return PT_OK;
}

int pt_renderer_reset_rng(pt_renderer* r) {
  if (!r) return pt_fail(PT_EINVAL, "pt_renderer_reset_rng: renderer is NULL");
  r->frame = 0;
  return setup_random(r);
}

int pt_renderer_get_rng_state(pt_renderer* r, uint32_t* h_state, size_t n_words) {
  if (!r || !h_state) return pt_fail(PT_EINVAL, "pt_renderer_get_rng_state: NULL argument");
  if (!r->d_state) return pt_fail(PT_EINVAL, "pt_renderer_get_rng_state: renderer keeps no generator state");
  if (n_words != (size_t)r->tile_pixels * 6) return pt_fail(PT_EINVAL, "pt_renderer_get_rng_state: expected %zu words", (size_t)r->tile_pixels * 6);
  PT_HIP(hipMemcpy(h_state, r->d_state, n_words * sizeof(uint32_t), hipMemcpyDeviceToHost));
  return PT_OK;
}

int pt_renderer_set_rng_state(pt_renderer* r, const uint32_t* h_state, size_t n_words) {
  if (!r || !h_state) return pt_fail(PT_EINVAL, "pt_renderer_set_rng_state: NULL argument");
  if (!r->d_state) return pt_fail(PT_EINVAL, "pt_renderer_set_rng_state: renderer keeps no generator state");
  if (n_words != (size_t)r->tile_pixels * 6) return pt_fail(PT_EINVAL, "pt_renderer_set_rng_state: expected %zu words", (size_t)r->tile_pixels * 6);
  PT_HIP(hipMemcpy(r->d_state, h_state, n_words * sizeof(uint32_t), hipMemcpyHostToDevice));
  return PT_OK;
}

int pt_renderer_kernel_info(pt_renderer* r, int n_spheres, pt_kernel_info* info) {
  if (!r || !info) return pt_fail(PT_EINVAL, "pt_renderer_kernel_info: NULL argument");
  hipFuncAttributes fa;
  const int variant = effective_variant(r, n_spheres);
  const bool fast = variant == PT_VARIANT_FAST;
  PT_HIP(hipFuncGetAttributes(&fa, fast ? pt_fast_kernel_symbol(r->opts.rng_mode, n_spheres, r->opts.max_bounces)
                                        : pt_kernel_symbol(r->opts.rng_mode, variant, n_spheres, r->opts.max_bounces,
                                                           r->opts.layout == PT_LAYOUT_PLANAR)));
  const int block = fast ? PT_BLOCK_THREADS : pt_kernel_block_threads(variant);
  info->block_threads = block;
  info->grid_blocks = (int)((r->tile_pixels + block - 1) / block);
  info->lds_bytes = (int)(fast ? pt_fast_kernel_lds_bytes(n_spheres) : pt_kernel_lds_bytes(n_spheres, variant));
  info->variant = variant;
  if (variant == 8 || variant == 9)
    info->grid_blocks = (int)(((uint64_t)r->tile_pixels * (variant == 8 ? 4 : 2) + PT_BLOCK_THREADS - 1) / PT_BLOCK_THREADS);
  if (!fast && chunks_of(r, variant) >= 2u &&
      pt_kernel_chunked(variant, n_spheres, r->opts.max_bounces, r->opts.layout == PT_LAYOUT_PLANAR, r->spp, chunks_of(r, variant)))
    info->grid_blocks *= (int)chunks_of(r, variant);  // sample chunking: that many workgroups per pixel block
  info->num_vgprs = fa.numRegs;
  info->reserved0 = 0;
  info->scratch_bytes = (int)fa.localSizeBytes;
  info->max_spheres = fast ? (1 << 26) : pt_kernel_max_spheres(variant);
  return PT_OK;
}

}  // extern "C"
