// pt_mgpu.hip -- one frame row-tiled over the GPUs of one node (include/ptcore.h, pt_mgpu_*).
//
// The reference is single-GPU: it selects one device for the whole process (src/main.cu:86) and its
// Renderer launches pixel_kernel over the whole image (include/Renderer.h:69).  Pixels are independent and
// every generator is keyed on the GLOBAL pixel id (src/pathtrace.cu:206,265), so the frame shards into
// contiguous row blocks, and because the buffer is [row][col][14] a row block is one contiguous span of it.
//
// Shape: ONE process, ONE host thread per device (each thread binds its device once and owns that device's
// renderer, stream and tile buffer), one exchange step per frame:
//   rank 0 (the root: the device that owns the caller's frame) posts one ncclRecv per peer STRAIGHT INTO the
//   frame at the peer's tile offset and renders its own tile in place; every other rank posts one ncclSend of
//   its tile -- all inside ncclGroupStart/End, i.e. what ncclGather does internally, but with zero-copy
//   placement and ragged tiles.  The G-1 transfers arrive over G-1 distinct xGMI links (a gather, not a ring:
//   the per-link ring bound does not apply).
// RCCL is loaded with dlopen on first use, so the single-GPU path of libptcore.so does not depend on it.
// Round 4: the exchange is pipelined INSIDE a frame.  A rank renders its tile as row bands (one launch each, one renderer each)
// and band b's send -- the root: band b's receives -- is posted on a second stream behind band b's kernel, so it travels while
// band b + 1 renders; only the last band's transfer is exposed.  Renderer::Render stays synchronous (include/Renderer.h:55-76).
// A second exchange backend, peer copies on the tile's stream (hipMemcpyPeerAsync: the SDMA engines move
// the tile over the same xGMI link, no CU involved), serves two ranks sharing one device (RCCL refuses
// duplicate devices in one communicator) and is selectable for A/B.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <dlfcn.h>

#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "pt_internal.h"

namespace {

// ---- RCCL entry points, resolved at run time -------------------------------------------------------
struct Rccl {
  void* handle = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*GetVersion)(int*) = nullptr;
  std::string error;
};

Rccl* load_rccl() {
  static std::mutex mu;
  static Rccl* lib = nullptr;
  std::lock_guard<std::mutex> lock(mu);
  if (lib) return lib;
  Rccl* r = new Rccl();
  // a copy already mapped into the process (e.g. by torch) is reused: two RCCL instances must not coexist
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char* n : names)
    if ((r->handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
  if (!r->handle)
    for (const char* n : names)
      if ((r->handle = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
  if (!r->handle) {
    r->error = std::string("cannot load librccl.so: ") + (dlerror() ? dlerror() : "?");
  } else {
#define PT_SYM(field, name)                                                         \
  r->field = reinterpret_cast<decltype(r->field)>(dlsym(r->handle, name));         \
  if (!r->field && r->error.empty()) r->error = std::string("librccl.so lacks ") + name;
    PT_SYM(CommInitAll, "ncclCommInitAll")
    PT_SYM(CommDestroy, "ncclCommDestroy")
    PT_SYM(CommAbort, "ncclCommAbort")
    PT_SYM(GroupStart, "ncclGroupStart")
    PT_SYM(GroupEnd, "ncclGroupEnd")
    PT_SYM(Send, "ncclSend")
    PT_SYM(Recv, "ncclRecv")
    PT_SYM(GetErrorString, "ncclGetErrorString")
    PT_SYM(GetVersion, "ncclGetVersion")
#undef PT_SYM
  }
  lib = r;
  return lib;
}

// contiguous, balanced row blocks: the first (height % world) ranks get one extra row
void row_range(int height, int world, int rank, int* begin, int* end) {
  const int base = height / world, extra = height % world;
  *begin = rank * base + (rank < extra ? rank : extra);
  *end = *begin + base + (rank < extra ? 1 : 0);
}

struct Job {  // one frame; filled by pt_mgpu_render, read by every worker
  float* d_out = nullptr;            // the caller's frame, on the root device
  const pt_sphere* d_spheres = nullptr;  // on the root device
  int n_spheres = 0;
  float basis[12];
  float eye[3];
};

}  // namespace

struct pt_mgpu {
  struct Rank {
    int rank = 0, device = 0, row_begin = 0, row_end = 0;
    // The tile is rendered as `bands` row bands, one renderer each (a renderer owns the generator state and the scratch of its
    // rows), launched back to back on `stream`; band b's part of the exchange is posted on `xfer` behind an event, so it travels
    // while band b + 1 renders and only the LAST band's transfer is exposed.  One band: one stream, as before round 4.
    std::vector<pt_renderer*> band;
    std::vector<int> band_begin;               // first row of band b (band_begin[bands] = row_end)
    std::vector<hipEvent_t> ev_band;           // band b has been rendered
    hipStream_t stream = nullptr, xfer = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;  // around this rank's kernels
    float* d_tile = nullptr;                   // rendered here unless the tile is rendered in place
    pt_sphere* d_scene = nullptr;              // this device's replica of the scene (non-root)
    int scene_capacity = 0;
    ncclComm_t comm = nullptr;
    std::thread thread;
    // per-frame results
    int rc = PT_OK;
    char err[400] = "";
    float kernel_ms = 0.0f;
  };
  int n = 0, width = 0, height = 0, spp = 0, tpb = 0;
  int bands = 1;             // row bands per tile (the same on every rank: band b of every peer is one grouped exchange step)
  float exposed_ms = 0.0f;   // last frame: wall time minus the longest rank's render time
  float render_ms = 0.0f;    // last frame: the longest rank's render time (first launch to last band done)
  pt_renderer_opts ropts;
  pt_mgpu_opts opts;
  bool use_rccl = false;
  Rccl* rccl = nullptr;
  std::vector<Rank> ranks;
  // dispatch: generation counter + condition variables
  std::mutex mu;
  std::condition_variable cv_go, cv_done;
  uint64_t generation = 0;
  int pending = 0;
  bool quit = false;
  bool init_phase = true;
  bool failed = false;  // a frame timed out and a communicator was aborted: the object only accepts destroy from then on
  Job job;
};

namespace {

#define W_HIP(call)                                                                                              \
  do {                                                                                                           \
    hipError_t e_ = (call);                                                                                      \
    if (e_ != hipSuccess) {                                                                                      \
      snprintf(rk.err, sizeof(rk.err), "rank %d (device %d): %s: %s", rk.rank, rk.device, #call, hipGetErrorString(e_)); \
      rk.rc = e_ == hipErrorNoDevice ? PT_ENODEVICE : PT_EHIP;                                                   \
      return;                                                                                                    \
    }                                                                                                            \
  } while (0)
#define W_PT(call)                                                                                   \
  do {                                                                                               \
    int rc_ = (call);                                                                                \
    if (rc_ != PT_OK) {                                                                              \
      snprintf(rk.err, sizeof(rk.err), "rank %d (device %d): %s", rk.rank, rk.device, pt_last_error()); \
      rk.rc = rc_;                                                                                   \
      return;                                                                                        \
    }                                                                                                \
  } while (0)
#define W_NCCL(call)                                                                                            \
  do {                                                                                                          \
    ncclResult_t r_ = (call);                                                                                   \
    if (r_ != ncclSuccess) {                                                                                    \
      snprintf(rk.err, sizeof(rk.err), "rank %d (device %d): %s: %s", rk.rank, rk.device, #call, m->rccl->GetErrorString(r_)); \
      rk.rc = PT_ECOMM;                                                                                         \
      return;                                                                                                   \
    }                                                                                                           \
  } while (0)

// the tile is rendered straight into the caller's frame when this rank IS the root (or shares its device)
// and nothing forces the exchange
bool in_place(const pt_mgpu* m, const pt_mgpu::Rank& rk) {
  if (m->opts.force_exchange) return false;
  return rk.device == m->ranks[0].device;
}

void worker_init(pt_mgpu* m, pt_mgpu::Rank& rk) {
  W_HIP(hipSetDevice(rk.device));
  W_HIP(hipStreamCreateWithFlags(&rk.stream, hipStreamNonBlocking));
  if (m->bands > 1) W_HIP(hipStreamCreateWithFlags(&rk.xfer, hipStreamNonBlocking));
  W_HIP(hipEventCreate(&rk.ev0));
  W_HIP(hipEventCreate(&rk.ev1));
  const int rows = rk.row_end - rk.row_begin;
  rk.band_begin.assign(m->bands + 1, rk.row_end);
  for (int b = 0; b < m->bands; b++) {
    int lo, hi;
    row_range(rows, m->bands, b, &lo, &hi);  // ragged like the tiles themselves; a tile of fewer rows than bands has empty ones
    rk.band_begin[b] = rk.row_begin + lo;
  }
  rk.band.assign(m->bands, nullptr);
  rk.ev_band.assign(m->bands, nullptr);
  for (int b = 0; b < m->bands; b++) {
    if (m->bands > 1) W_HIP(hipEventCreateWithFlags(&rk.ev_band[b], hipEventDisableTiming));
    pt_renderer_opts o = m->ropts;
    o.row_begin = rk.band_begin[b];
    o.row_end = rk.band_begin[b + 1];
    if (o.row_end > o.row_begin) W_PT(pt_renderer_create(m->width, m->height, m->spp, m->tpb, &o, &rk.band[b]));
  }
  if (rows > 0 && !in_place(m, rk)) W_HIP(hipMalloc((void**)&rk.d_tile, (size_t)rows * m->width * 14 * sizeof(float)));
}

#if PT_BUILD_EXPERIMENTS
// lab library: PT_LAB_MGPU_STALL="<rank>:<ms>" holds that rank's stream (a host function that sleeps) between its render and
// its part of the exchange step, so the frame misses its deadline -- the only way to reach the PT_ETIMEOUT / ncclCommAbort
// path on a machine where every peer is healthy (tests/test_mgpu_gpu.py)
void lab_sleep_ms(void* ms) { std::this_thread::sleep_for(std::chrono::milliseconds((long)(intptr_t)ms)); }
#endif

// One frame of one rank.  Errors before or inside the exchange step are REMEMBERED, not returned at once: a rank that failed
// to render still posts its part of the grouped send/recv (the root would otherwise wait out the whole deadline for a tile
// that never comes), and a group that was started is always ended.
void worker_frame(pt_mgpu* m, pt_mgpu::Rank& rk) {
  const Job& j = m->job;
  const pt_mgpu::Rank& root = m->ranks[0];
  const size_t row_floats = (size_t)m->width * 14;
  const size_t count = (size_t)(rk.row_end - rk.row_begin) * row_floats;
  rk.kernel_ms = 0.0f;
  int first_rc = PT_OK;  // the first failure of this frame; reported when the rank is done
  auto note_hip = [&](hipError_t e, const char* what) {
    if (e == hipSuccess) return true;
    if (first_rc == PT_OK) {
      snprintf(rk.err, sizeof(rk.err), "rank %d (device %d): %s: %s", rk.rank, rk.device, what, hipGetErrorString(e));
      first_rc = e == hipErrorNoDevice ? PT_ENODEVICE : PT_EHIP;
    }
    return false;
  };
  auto note_pt = [&](int rc) {
    if (rc == PT_OK) return true;
    if (first_rc == PT_OK) {
      snprintf(rk.err, sizeof(rk.err), "rank %d (device %d): %s", rk.rank, rk.device, pt_last_error());
      first_rc = rc;
    }
    return false;
  };
  auto note_nccl = [&](ncclResult_t r, const char* what) {
    if (r == ncclSuccess) return true;
    if (first_rc == PT_OK) {
      snprintf(rk.err, sizeof(rk.err), "rank %d (device %d): %s: %s", rk.rank, rk.device, what, m->rccl->GetErrorString(r));
      first_rc = PT_ECOMM;
    }
    return false;
  };
  // scene replica: the caller's spheres live on the root device (Scene::objects); 360 B .. 40 KB per frame
  const pt_sphere* scene = j.d_spheres;
  bool scene_ok = true;
  if (rk.device != root.device && j.n_spheres > 0 && count) {
    if (rk.scene_capacity < j.n_spheres) {
      if (rk.d_scene) (void)hipFree(rk.d_scene);
      rk.d_scene = nullptr;
      rk.scene_capacity = 0;
      scene_ok = note_hip(hipMalloc((void**)&rk.d_scene, (size_t)j.n_spheres * sizeof(pt_sphere)), "hipMalloc(scene replica)");
      if (scene_ok) rk.scene_capacity = j.n_spheres;
    }
    if (scene_ok)
      scene_ok = note_hip(hipMemcpyPeerAsync(rk.d_scene, rk.device, j.d_spheres, root.device, (size_t)j.n_spheres * sizeof(pt_sphere), rk.stream),
                          "hipMemcpyPeerAsync(scene replica)");
    scene = rk.d_scene;
  }
  float* target = in_place(m, rk) ? j.d_out + (size_t)rk.row_begin * row_floats : rk.d_tile;
  hipStream_t xs = m->bands > 1 ? rk.xfer : rk.stream;  // where the exchange is posted
  bool timed = false, comm_broken = false;
  if (count && scene_ok) timed = note_hip(hipEventRecord(rk.ev0, rk.stream), "hipEventRecord");
  if (m->use_rccl && !rk.comm) comm_broken = true;
  for (int b = 0; b < m->bands; b++) {
    const int b0 = rk.band_begin[b], b1 = rk.band_begin[b + 1];
    const size_t bcount = (size_t)(b1 - b0) * row_floats, boff = (size_t)(b0 - rk.row_begin) * row_floats;
    // ---- render band b ----
    bool rendered = false;
    if (bcount && scene_ok && rk.band[b])
      rendered = note_pt(pt_renderer_enqueue(rk.band[b], target + boff, scene, j.n_spheres, j.basis, j.eye, rk.stream));
    (void)rendered;
#if PT_BUILD_EXPERIMENTS
    if (b == m->bands - 1)
      if (const char* st = getenv("PT_LAB_MGPU_STALL")) {
        int r = -1, ms = 0;
        if (sscanf(st, "%d:%d", &r, &ms) == 2 && r == rk.rank && ms > 0)
          note_hip(hipLaunchHostFunc(rk.stream, lab_sleep_ms, (void*)(intptr_t)ms), "hipLaunchHostFunc");
      }
#endif
    if (m->bands > 1) {  // band b's exchange waits for band b's kernel only
      note_hip(hipEventRecord(rk.ev_band[b], rk.stream), "hipEventRecord");
      note_hip(hipStreamWaitEvent(xs, rk.ev_band[b], 0), "hipStreamWaitEvent");
    }
    // ---- band b's exchange step: posted even by a rank whose render failed (the frame fails, but nobody waits for it) -----
    if (m->use_rccl) {
      if (comm_broken) continue;
      if (note_nccl(m->rccl->GroupStart(), "ncclGroupStart")) {
        if (rk.rank == 0) {
          for (const pt_mgpu::Rank& p : m->ranks) {
            const size_t pc = (size_t)(p.band_begin[b + 1] - p.band_begin[b]) * row_floats;
            if (pc && !in_place(m, p))
              comm_broken |= !note_nccl(m->rccl->Recv(j.d_out + (size_t)p.band_begin[b] * row_floats, pc, ncclFloat, p.rank, rk.comm, xs), "ncclRecv");
          }
        }
        if (bcount && !in_place(m, rk)) comm_broken |= !note_nccl(m->rccl->Send(rk.d_tile + boff, bcount, ncclFloat, 0, rk.comm, xs), "ncclSend");
        comm_broken |= !note_nccl(m->rccl->GroupEnd(), "ncclGroupEnd");  // a started group is always ended
      } else {
        comm_broken = true;
      }
    } else if (bcount && !in_place(m, rk)) {
      note_hip(hipMemcpyPeerAsync(j.d_out + (size_t)b0 * row_floats, root.device, rk.d_tile + boff, rk.device, bcount * sizeof(float), xs),
               "hipMemcpyPeerAsync(band)");
    }
  }
  if (count && scene_ok) timed = note_hip(hipEventRecord(rk.ev1, rk.stream), "hipEventRecord") && timed;
  if (m->use_rccl && comm_broken && rk.comm) {  // peers may be waiting for operations this rank could not post: unblock them
    (void)m->rccl->CommAbort(rk.comm);
    rk.comm = nullptr;
  }
  // ---- completion, with a deadline: a peer that never arrives must not hang the caller ----------------
  const auto deadline = std::chrono::steady_clock::now() + std::chrono::milliseconds(m->opts.timeout_ms);
  for (;;) {
    hipError_t q = hipStreamQuery(rk.stream);
    if (q == hipSuccess && m->bands > 1) q = hipStreamQuery(rk.xfer);
    if (q == hipSuccess) break;
    if (q != hipErrorNotReady) {
      note_hip(q, "hipStreamQuery");
      break;
    }
    if (m->opts.timeout_ms > 0 && std::chrono::steady_clock::now() > deadline) {
      if (m->use_rccl && rk.comm) {
        (void)m->rccl->CommAbort(rk.comm);  // unblocks the device-side wait of the grouped send/recv
        rk.comm = nullptr;
      }
      if (first_rc == PT_OK)
        snprintf(rk.err, sizeof(rk.err), "rank %d (device %d): frame not complete after %d ms (exchange aborted)", rk.rank, rk.device,
                 m->opts.timeout_ms);
      rk.rc = PT_ETIMEOUT;  // outranks whatever else went wrong: the object is unusable from here on
      return;
    }
    std::this_thread::sleep_for(std::chrono::microseconds(20));  // a frame is >= milliseconds; do not burn a core per rank
  }
  // the stream has drained: a frame whose sample-chunk chain broke belongs to THIS call, not to the next one (or to nobody)
  for (pt_renderer* r : rk.band)
    if (r) note_pt(pt_renderer_check(r, 1, nullptr));
  if (timed && first_rc == PT_OK) note_hip(hipEventElapsedTime(&rk.kernel_ms, rk.ev0, rk.ev1), "hipEventElapsedTime");
  rk.rc = first_rc;
}

void worker_main(pt_mgpu* m, int index) {
  pt_mgpu::Rank& rk = m->ranks[index];
  uint64_t seen = 0;
  for (;;) {
    bool init;
    {
      std::unique_lock<std::mutex> lock(m->mu);
      m->cv_go.wait(lock, [&] { return m->quit || m->generation != seen; });
      if (m->quit) break;
      seen = m->generation;
      init = m->init_phase;
    }
    rk.rc = PT_OK;
    rk.err[0] = 0;
    if (init) worker_init(m, rk); else worker_frame(m, rk);
    {
      std::lock_guard<std::mutex> lock(m->mu);
      if (--m->pending == 0) m->cv_done.notify_all();
    }
  }
  // teardown on the owning thread (the device binding is per thread)
  (void)hipSetDevice(rk.device);
  for (pt_renderer* r : rk.band)
    if (r) (void)pt_renderer_destroy(r);
  for (hipEvent_t e : rk.ev_band)
    if (e) (void)hipEventDestroy(e);
  if (rk.xfer) (void)hipStreamDestroy(rk.xfer);
  if (rk.d_tile) (void)hipFree(rk.d_tile);
  if (rk.d_scene) (void)hipFree(rk.d_scene);
  if (rk.ev0) (void)hipEventDestroy(rk.ev0);
  if (rk.ev1) (void)hipEventDestroy(rk.ev1);
  if (rk.stream) (void)hipStreamDestroy(rk.stream);
}

// run one generation on every worker and collect the first failure
int run_all(pt_mgpu* m, const char* what) {
  {
    std::unique_lock<std::mutex> lock(m->mu);
    m->pending = m->n;
    m->generation++;
    m->cv_go.notify_all();
    m->cv_done.wait(lock, [&] { return m->pending == 0; });
  }
  // a timeout outranks every other failure (it is the one that makes the object unusable), then the first failing rank
  for (const pt_mgpu::Rank& rk : m->ranks)
    if (rk.rc == PT_ETIMEOUT) return pt_fail(rk.rc, "%s: %s", what, rk.err);
  for (const pt_mgpu::Rank& rk : m->ranks)
    if (rk.rc != PT_OK) return pt_fail(rk.rc, "%s: %s", what, rk.err);
  return PT_OK;
}

int env_int(const char* name, int fallback) {
  const char* v = getenv(name);
  return (v && *v) ? atoi(v) : fallback;
}

}  // namespace

extern "C" {

void pt_mgpu_opts_default(pt_mgpu_opts* o) {
  if (!o) return;
  memset(o, 0, sizeof(*o));
  o->gather = PT_GATHER_AUTO;
  o->force_exchange = env_int("PT_FORCE_MGPU", 0) ? 1 : 0;
  o->timeout_ms = env_int("PT_MGPU_TIMEOUT_MS", 60000);
  o->bands = env_int("PT_MGPU_BANDS", 0);
  const char* g = getenv("PT_MGPU_GATHER");
  if (g && !strcmp(g, "rccl")) o->gather = PT_GATHER_RCCL;
  if (g && !strcmp(g, "copy")) o->gather = PT_GATHER_PEER_COPY;
}

int pt_mgpu_create(int n_gpus, const int* devices, int width, int height, int samples_per_pixel, int threads_per_block,
                   const pt_renderer_opts* opts, const pt_mgpu_opts* mopts, pt_mgpu** out) {
  if (!out) return pt_fail(PT_EINVAL, "pt_mgpu_create: out is NULL");
  *out = nullptr;
  if (n_gpus < 1 || n_gpus > 64) return pt_fail(PT_EINVAL, "pt_mgpu_create: n_gpus %d", n_gpus);
  if (width <= 0 || height <= 0 || samples_per_pixel <= 0)
    return pt_fail(PT_EINVAL, "pt_mgpu_create: width/height/samples must be positive (%d x %d x %d)", width, height, samples_per_pixel);
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0) return pt_fail(PT_ENODEVICE, "pt_mgpu_create: no HIP device (%s)", hipGetErrorString(e));
  pt_mgpu* m = new (std::nothrow) pt_mgpu();
  if (!m) return pt_fail(PT_ENOMEM, "pt_mgpu_create: out of host memory");
  m->n = n_gpus;
  m->width = width;
  m->height = height;
  m->spp = samples_per_pixel;
  m->tpb = threads_per_block;
  if (opts) m->ropts = *opts; else pt_renderer_opts_default(&m->ropts);
  if (m->ropts.row_begin != 0 || m->ropts.row_end != 0) {
    delete m;
    return pt_fail(PT_EINVAL, "pt_mgpu_create: opts.row_begin/row_end must be 0 (the tiles are chosen here)");
  }
  if (mopts) m->opts = *mopts; else pt_mgpu_opts_default(&m->opts);
  m->ranks.resize(n_gpus);
  bool duplicates = false;
  for (int g = 0; g < n_gpus; g++) {
    pt_mgpu::Rank& rk = m->ranks[g];
    rk.rank = g;
    rk.device = devices ? devices[g] : g;
    if (rk.device < 0 || rk.device >= ndev) {
      const int bad = rk.device;
      delete m;
      return pt_fail(PT_EINVAL, "pt_mgpu_create: device %d of rank %d does not exist (%d visible)", bad, g, ndev);
    }
    for (int k = 0; k < g; k++) duplicates |= m->ranks[k].device == rk.device;
    row_range(height, n_gpus, g, &rk.row_begin, &rk.row_end);
  }
  // exchange backend: RCCL between distinct devices; peer copies when ranks share a device (RCCL refuses that)
  const bool anything_to_exchange = n_gpus > 1 || m->opts.force_exchange;
  if (m->opts.gather == PT_GATHER_RCCL && duplicates) {
    delete m;
    return pt_fail(PT_EINVAL, "pt_mgpu_create: PT_GATHER_RCCL needs distinct devices");
  }
  m->use_rccl = anything_to_exchange && (m->opts.gather == PT_GATHER_RCCL || (m->opts.gather == PT_GATHER_AUTO && !duplicates));
  // Row bands per tile.  A band must still fill the chip -- at least eight one-lane waves per SIMD (BASELINE configs[2]: a
  // rank's 512 x 4096 tile = 32 waves per SIMD -> 4 bands of 128 rows; configs[1]: a rank's tile is 2 waves per SIMD -> 1 band,
  // and its 7 MB are not worth hiding) -- and there is nothing to hide unless a tile crosses a link: an exchange between ranks
  // that share the root's device is an on-device copy at HBM speed (117 MB in 0.05-0.1 ms), against which four launches instead
  // of one cost the render 2 % (profiles/r04/mgpu_bands.txt).  opts.bands > 0 bands whatever the devices are.
  if (m->opts.bands < 0 || m->opts.bands > 64) {
    const int bad = m->opts.bands;
    delete m;
    return pt_fail(PT_EINVAL, "pt_mgpu_create: bands %d (0 = automatic, 1..64)", bad);
  }
  m->bands = 1;
  if (anything_to_exchange) {
    if (m->opts.bands > 0) {
      m->bands = m->opts.bands;
    } else {
      bool crosses_link = false;
      for (const pt_mgpu::Rank& rk : m->ranks) crosses_link |= rk.device != m->ranks[0].device;
      hipDeviceProp_t prop;
      if (crosses_link && hipGetDeviceProperties(&prop, m->ranks[0].device) == hipSuccess && prop.multiProcessorCount > 0) {
        const double tile_pixels = (double)(m->ranks[0].row_end - m->ranks[0].row_begin) * width;
        const int b = (int)(tile_pixels / ((double)prop.multiProcessorCount * 4.0 * 64.0) / 8.0);
        m->bands = b < 1 ? 1 : (b > 8 ? 8 : b);
      }
    }
  }
  if (m->use_rccl) {
    m->rccl = load_rccl();
    if (!m->rccl->error.empty()) {
      std::string msg = m->rccl->error;
      delete m;
      return pt_fail(PT_ECOMM, "pt_mgpu_create: %s", msg.c_str());
    }
    std::vector<int> devs(n_gpus);
    std::vector<ncclComm_t> comms(n_gpus, nullptr);
    for (int g = 0; g < n_gpus; g++) devs[g] = m->ranks[g].device;
    ncclResult_t r = m->rccl->CommInitAll(comms.data(), n_gpus, devs.data());  // rccl.h:236
    if (r != ncclSuccess) {
      const char* s = m->rccl->GetErrorString(r);
      delete m;
      return pt_fail(PT_ECOMM, "pt_mgpu_create: ncclCommInitAll over %d devices: %s", n_gpus, s);
    }
    for (int g = 0; g < n_gpus; g++) m->ranks[g].comm = comms[g];
  }
  for (int g = 0; g < n_gpus; g++) m->ranks[g].thread = std::thread(worker_main, m, g);
  m->init_phase = true;
  int rc = run_all(m, "pt_mgpu_create");
  m->init_phase = false;
  if (rc != PT_OK) {
    char keep[512];
    snprintf(keep, sizeof(keep), "%s", pt_last_error());
    pt_mgpu_destroy(m);
    return pt_fail(rc, "%s", keep);
  }
  *out = m;
  return PT_OK;
}

int pt_mgpu_destroy(pt_mgpu* m) {
  if (!m) return PT_OK;
  {
    std::lock_guard<std::mutex> lock(m->mu);
    m->quit = true;
    m->cv_go.notify_all();
  }
  for (pt_mgpu::Rank& rk : m->ranks)
    if (rk.thread.joinable()) rk.thread.join();
  if (m->rccl)
    for (pt_mgpu::Rank& rk : m->ranks)
      if (rk.comm) (void)m->rccl->CommDestroy(rk.comm);
  delete m;
  return PT_OK;
}

int pt_mgpu_render(pt_mgpu* m, float* d_out, const pt_sphere* d_spheres, int n_spheres, const float basis[12],
                   const float eye[3], float* ms_out) {
  if (!m) return pt_fail(PT_EINVAL, "pt_mgpu_render: handle is NULL");
  if (!d_out) return pt_fail(PT_EINVAL, "pt_mgpu_render: d_out is NULL");
  if (n_spheres < 0 || (n_spheres > 0 && !d_spheres)) return pt_fail(PT_EINVAL, "pt_mgpu_render: bad scene (%d spheres)", n_spheres);
  if (!basis || !eye) return pt_fail(PT_EINVAL, "pt_mgpu_render: basis/eye is NULL");
  if (m->failed)
    return pt_fail(PT_ECOMM, "pt_mgpu_render: an earlier frame timed out or lost its communicator (exchange aborted); the only call this "
                             "object still accepts is pt_mgpu_destroy -- destroy and re-create");
  m->job.d_out = d_out;
  m->job.d_spheres = d_spheres;
  m->job.n_spheres = n_spheres;
  memcpy(m->job.basis, basis, sizeof(m->job.basis));
  memcpy(m->job.eye, eye, sizeof(m->job.eye));
  const auto t0 = std::chrono::steady_clock::now();
  const int rc = run_all(m, "pt_mgpu_render");
  // unusable from here on if ANY rank timed out or has aborted its communicator, whichever rank's error is reported
  for (const pt_mgpu::Rank& rk : m->ranks)
    if (rk.rc == PT_ETIMEOUT || (m->use_rccl && !rk.comm)) m->failed = true;
  const float wall = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
  if (ms_out) *ms_out = wall;
  m->render_ms = 0.0f;
  for (const pt_mgpu::Rank& rk : m->ranks) m->render_ms = rk.kernel_ms > m->render_ms ? rk.kernel_ms : m->render_ms;
  m->exposed_ms = wall > m->render_ms ? wall - m->render_ms : 0.0f;
  return rc;
}

int pt_mgpu_frame_stats(pt_mgpu* m, int* bands, float* render_ms, float* exposed_ms) {
  if (!m) return pt_fail(PT_EINVAL, "pt_mgpu_frame_stats: handle is NULL");
  if (bands) *bands = m->bands;
  if (render_ms) *render_ms = m->render_ms;
  if (exposed_ms) *exposed_ms = m->exposed_ms;
  return PT_OK;
}

int pt_mgpu_tile(pt_mgpu* m, int rank, int* device, int* row_begin, int* row_end, float* kernel_ms) {
  if (!m || rank < 0 || rank >= m->n) return pt_fail(PT_EINVAL, "pt_mgpu_tile: bad handle or rank");
  const pt_mgpu::Rank& rk = m->ranks[rank];
  if (device) *device = rk.device;
  if (row_begin) *row_begin = rk.row_begin;
  if (row_end) *row_end = rk.row_end;
  if (kernel_ms) *kernel_ms = rk.kernel_ms;
  return PT_OK;
}

int pt_mgpu_backend(pt_mgpu* m, char* name, size_t name_len) {
  if (!m || !name || !name_len) return pt_fail(PT_EINVAL, "pt_mgpu_backend: NULL argument");
  if (m->use_rccl) {
    int v = 0;
    (void)m->rccl->GetVersion(&v);
    snprintf(name, name_len, "rccl %d.%d.%d grouped send/recv", v / 10000, (v / 100) % 100, v % 100);
  } else if (m->n > 1 || m->opts.force_exchange) {
    snprintf(name, name_len, "hipMemcpyPeerAsync");
  } else {
    snprintf(name, name_len, "none (single tile rendered in place)");
  }
  return PT_OK;
}

}  // extern "C"
