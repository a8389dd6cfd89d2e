#!/usr/bin/env python3
"""bench.py -- headline benchmark of the path-trace megakernel on MI355X.

Metric (BASELINE.json): Msamples/s (+ ms/frame) on the 9-sphere Cornell box, 1024 x 1024,
1024 spp, fixed seed (--config cfg2, the default).  --config cfg3 is BASELINE.json configs[2]:
4096 x 4096 x 64 spp, the 8-GPU row-tiled configuration; cfg4 / cfg4open are configs[3] (1000 spheres,
1024 x 1024 x 256 spp, closed and open) and cfg5 is configs[4] (512 x 512 x 4 spp x 8 bounces, one frame per
step, generator state carried from frame to frame) -- the reference prints its time for any --size/-s
(src/main.cu:183), so every configuration has a bench line.  The default single-GPU run also measures those
four after the headline (`other_configs`: kernel ms + Msamples/s).  A "step" is one frame: one pass of the
hot path over the whole image.  At N GPUs the image is row-tiled (rank g renders rows
row_range(H, N, g)) and gathered to rank 0 over RCCL at frame end; the timed region includes
that gather.  Total work is fixed as N grows -> "scaling": "strong".

  python bench.py --gpus 1 --steps 5 --warmup 1
  python bench.py --gpus N [--config cfg3]        # spawns its N ranks itself (one process per GPU, RCCL)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
      --master-port P bench.py --gpus N --steps K --warmup W      # the driver's form: same ranks
  python bench.py --gpus N --engine native       # ONE process: libptcore's pt_mgpu_* (a host thread per
                                                 # device, RCCL inside the library), no torch.distributed

Rank 0 prints ONE JSON line.  Inputs (scene 360 B, camera 60 B) are resident / kernel
arguments before the timed region starts; output stays in HBM (the reference's interactive
mode never copies it to the host either, src/main.cu:146-177).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {  # BASELINE.json `configs`
    "cfg2": {"width": 1024, "height": 1024, "spp": 1024, "name": "BASELINE.json configs[1]", "scene": "cornell", "max_bounces": 5},
    "cfg3": {"width": 4096, "height": 4096, "spp": 64, "name": "BASELINE.json configs[2]", "scene": "cornell", "max_bounces": 5},
    # configs[3]: 1000 seeded random spheres inside the box, closed (6 wall spheres + light) and open (no walls: rays escape)
    "cfg4": {"width": 1024, "height": 1024, "spp": 256, "name": "BASELINE.json configs[3], closed", "scene": "random1000_walls", "max_bounces": 5},
    "cfg4open": {"width": 1024, "height": 1024, "spp": 256, "name": "BASELINE.json configs[3], open", "scene": "random1000_open", "max_bounces": 5},
    # configs[4]: the interactive shape -- one step = one 512 x 512 x 4 spp frame, 8 bounces, generator state carried over
    "cfg5": {"width": 512, "height": 512, "spp": 4, "name": "BASELINE.json configs[4], one frame per step", "scene": "cornell", "max_bounces": 8},
}


def scene_of(pt, cfg):
    if cfg["scene"] == "cornell":
        return pt.scene_cornell(), "9-sphere Cornell box"
    walls = cfg["scene"].endswith("walls")
    return pt.scene_random(1000, seed=1, with_walls=walls), ("1000 random spheres (seed 1) " + ("incl. 6 walls + light" if walls else "without walls"))
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
BYTES_PER_PIXEL = 56   # 14 x f32 written per pixel per frame
# VALU issue peak from MI355X_MICROARCH.md: a wave64 VALU instruction occupies its SIMD-32 for 2 cycles
# ("v_fma_f32 (wave64): 2 cyc"), 256 CUs x 4 SIMDs, 2.4 GHz -> 1024 x 2.4e9 / 2 wave-instructions per second
VALU_PEAK_GINST = 256 * 4 * 2.4 / 2.0  # = 1228.8 G wave-instr/s


def usable_cores():
    """Threads the CPU baseline may really use: the scheduler affinity mask, capped by the
    cgroup CPU quota (a GPU box exposes all 256 logical CPUs but grants a 16-CPU share)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


def cpu_baseline(oracle, cfg, spheres, basis, rows):
    """Oracle (CPU restatement, kind 'port') on the usable host cores over a bounded sample of the
    same workload: a band of `rows` full-width image rows at the configuration's spp.  Two builds of
    the same source: the parity build (-O2 -ffp-contract=off: the arithmetic the GPU
    is checked against; this is `value`) and -O3 -march=native (`native_build`, timing only)."""
    w, h, spp = cfg["width"], cfg["height"], cfg["spp"]
    cores = usable_cores()
    mb = cfg.get("max_bounces", 5)
    reps = 1  # a configuration whose whole frame is a fraction of a second of CPU work (cfg5) is rendered several times
    if rows <= 0:  # size the sample for about 8 s of wall time per build from a short probe, capped at the full frame
        probe_rows = max(1, min(h - h // 2, 8 * 1024 * 1024 * 9 // (w * spp * max(9, len(spheres)))))  # (a sample costs one test per sphere and bounce)
        t = time.perf_counter()
        oracle.render(w, h, spp, spheres=spheres, basis=basis, row_begin=h // 2, row_end=h // 2 + probe_rows, threads=cores, max_bounces=mb)
        rate = probe_rows * w * spp / (time.perf_counter() - t)
        rows = int(max(probe_rows, min(h, 8.0 * rate / (w * spp))))
        if rows >= h:
            reps = int(max(1, min(1000, 8.0 * rate / (w * h * spp))))
    rows = min(rows, h)
    r0 = h // 2 - rows // 2
    samples = rows * w * spp * reps

    def timed(native):
        t = time.perf_counter()
        for _ in range(reps):
            oracle.render(w, h, spp, spheres=spheres, basis=basis, row_begin=r0, row_end=r0 + rows, threads=cores, native=native, max_bounces=mb)
        return time.perf_counter() - t

    dt = timed(False)
    out = {
        "value": round(samples / dt / 1e6, 3),
        "unit": "Msamples/s",
        "cores": cores,
        "kind": "port",
        "sample": f"rows {r0}..{r0 + rows - 1} of the {w}x{h} frame at {spp} spp" + (f", rendered {reps} times" if reps > 1 else "") +
                  f" ({samples / 1e6:.1f} Msamples, {dt:.1f} s wall), "
                  "gcc -O2 -ffp-contract=off (the parity build), pthreads over rows",
    }
    try:
        oracle.native_lib()  # compiled here, on the machine it runs on
        dn = timed(True)
        out["native_build"] = {"value": round(samples / dn / 1e6, 3), "unit": "Msamples/s", "cores": cores,
                               "flags": "gcc -O3 -march=native (default FP contraction; timing only, not the contract's arithmetic)",
                               "seconds": round(dn, 1)}
    except Exception as e:  # a missing compiler must not lose the headline
        out["native_build"] = {"error": str(e)[:200]}
    return out


def spawn_ranks(args):
    """`python bench.py --gpus N` outside torch.distributed.run: start the N ranks as fresh child processes
    (this parent has not touched the GPU and never does), one per GPU, rendezvous on 127.0.0.1."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(args.gpus):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                r = p.poll()
                if r is None:
                    continue
                pending.remove(p)
                if r != 0 and rc == 0:
                    rc = r
                    for q in pending:  # one rank failed: the others would wait in a collective forever
                        q.terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


def profile_records(pt, cfg_key, rng, ki):
    """Counters measured with rocprofv3 (profiles/*.json) for exactly this build and kernel, or (None, None, why)."""
    key = f"{cfg_key}_{rng}_v{ki['variant']}"
    why = None
    recs = []
    for name in ("hbm_traffic.json", "valu_roofline.json"):
        path = os.path.join(ROOT, "profiles", name)
        rec = None
        if os.path.exists(path):
            try:
                rec = json.load(open(path)).get(key)
            except Exception:
                rec = None
        if rec is None:
            missing = f"no profile for {key}"
            why = why or (missing if name == "valu_roofline.json" or not os.path.exists(path) else None)  # (a profile may hold the VALU counters only)
        elif rec.get("fingerprint") != pt.build_fingerprint() or rec.get("num_vgprs") != ki["num_vgprs"]:
            why = f"profile {key} was measured on build {rec.get('fingerprint')} ({rec.get('num_vgprs')} VGPRs), this is {pt.build_fingerprint()} ({ki['num_vgprs']} VGPRs)"
            rec = None
        recs.append(rec)
    return recs[0], recs[1], why


def other_configs(pt, torch, device, stream, rng_mode):
    """Side record of the default run: the BASELINE.json configurations other than the headline on this one GPU --
    cfg3 (4096^2 x 64 spp), cfg4 closed / open (1000 spheres, 1024^2 x 256 spp), and cfg5 as a stream of 200 frames
    (512^2 x 4 spp x 8 bounces, enqueued back to back, one synchronisation at the end).  Kernel milliseconds are measured
    with events on the launch stream; about two seconds of GPU time in total."""
    out = {}
    for key, warm, reps in (("cfg3", 1, 2), ("cfg4", 1, 2), ("cfg4open", 1, 2), ("cfg5", 5, 200)):
        c = CONFIGS[key]
        w, h, spp, mb = c["width"], c["height"], c["spp"], c["max_bounces"]
        sph, label = scene_of(pt, c)
        d_scene = torch.from_numpy(sph.view("u1").reshape(-1).copy()).to(device)
        frame = torch.empty(h * w * 14, dtype=torch.float32, device=device)
        basis = pt.camera_basis(width=w, height=h)
        r = pt.Renderer(w, h, spp, rng_mode=rng_mode, persist_rng=True, max_bounces=mb)
        for _ in range(warm):
            r.enqueue(frame.data_ptr(), d_scene.data_ptr(), len(sph), basis, pt.DEFAULT_EYE, stream=stream.cuda_stream)
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(min(reps, 20))]
        t0 = time.perf_counter()
        for k in range(reps):
            if k < len(ev):
                ev[k][0].record(stream)
            r.enqueue(frame.data_ptr(), d_scene.data_ptr(), len(sph), basis, pt.DEFAULT_EYE, stream=stream.cuda_stream)
            if k < len(ev):
                ev[k][1].record(stream)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        kms = sorted(a.elapsed_time(b) for a, b in ev)
        ki = r.kernel_info(len(sph))
        display = None
        if key == "cfg5":
            # the reference's loop body is Render THEN Denoise (main.cu:148,175): the same stream of frames with the display pack
            # as a second launch per frame, and with the pack fused into the render kernel (pt_renderer_set_display)
            vtx = torch.empty(h * w * 3, dtype=torch.float32, device=device)
            # Protocol (one 200-frame window per mode could not tell a host stall from a slow kernel): the two modes
            # ALTERNATE, five windows of 1000 frames each, one perf_counter pair per window; min / median / max over the windows
            # are reported per mode, and beside them the KERNEL milliseconds of 20 frames per window from events on the launch
            # stream (render, or render + pack).  A stalled host shows as max >> median with the kernel time unchanged.
            n_win, n_frames, n_ev = 5, 1000, 20
            walls = {"separate": [], "fused": []}
            kern = {"separate": [], "fused": []}

            def one_frame(mode):
                r.enqueue(frame.data_ptr(), d_scene.data_ptr(), len(sph), basis, pt.DEFAULT_EYE, stream=stream.cuda_stream)
                if mode == "separate":
                    pt.check(pt.lib.pt_display_pack(frame.data_ptr(), w, h, vtx.data_ptr(), stream.cuda_stream))

            for win in range(n_win):
                for mode in ("separate", "fused"):
                    r.set_display(vtx.data_ptr() if mode == "fused" else None)
                    for _ in range(warm):
                        one_frame(mode)
                    torch.cuda.synchronize()
                    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_ev)]
                    t1 = time.perf_counter()
                    for k in range(n_frames):
                        sample = k % (n_frames // n_ev) == 0 and k // (n_frames // n_ev) < n_ev
                        if sample:
                            evs[k // (n_frames // n_ev)][0].record(stream)
                        one_frame(mode)
                        if sample:
                            evs[k // (n_frames // n_ev)][1].record(stream)
                    torch.cuda.synchronize()
                    walls[mode].append((time.perf_counter() - t1) / n_frames * 1e3)
                    kern[mode] += [a.elapsed_time(b) for a, b in evs]
            r.set_display(None)

            def mmm(v):
                v = sorted(v)
                return {"min": round(v[0], 4), "median": round(v[len(v) // 2], 4), "max": round(v[-1], 4)}

            display = {"render_plus_pack_launch_ms_per_frame": mmm(walls["separate"])["median"],
                       "render_with_fused_pack_ms_per_frame": mmm(walls["fused"])["median"],
                       "windows": {"count": n_win, "frames_each": n_frames, "order": "separate, fused alternating"},
                       "wall_ms_per_frame": {"separate": mmm(walls["separate"]), "fused": mmm(walls["fused"])},
                       "kernel_ms_per_frame": {"separate": mmm(kern["separate"]), "fused": mmm(kern["fused"])},
                       "note": "Denoiser::Denoise (display vertices, 12 B per pixel) after every frame (main.cu:148,175): as its own launch, and "
                               "fused into the render kernel's epilogue (pt_renderer_set_display); wall ms per frame in streams of frames, "
                               "kernel ms from events around sampled frames (separate: render + pack)"}
            del vtx
            # the same frames when their cameras are known in advance (a scripted fly-through, a pose sweep): one launch per 32
            # frames (pt_renderer_enqueue_frames) against the stream of single-frame launches, both into 256 separate frame buffers
            import numpy as np

            nb = 256
            big = torch.empty(nb * h * w * 14, dtype=torch.float32, device=device)
            bases = np.tile(np.asarray(basis, dtype=np.float32).reshape(1, 12), (nb, 1))
            eyes = np.tile(np.asarray(pt.DEFAULT_EYE, dtype=np.float32).reshape(1, 3), (nb, 1))
            frames_batch = {"frames": nb, "windows": 5,
                            "note": "pt_renderer_enqueue_frames, one launch per 32 frames, against the stream of single-frame launches, both into 256 "
                                    "separate frame buffers; wall ms per frame.  xorwow: a workgroup keeps its pixels and loops over the frames, the "
                                    "generator stays in registers; philox (re-keyed per frame): one workgroup per (frame, pixel block).  Every frame "
                                    "bit-identical to the single-frame path (tests/test_frames_gpu.py)"}
            for gen_name, gen in (("xorwow", pt.RNG_XORWOW), ("philox", pt.RNG_PHILOX)):
                rb = pt.Renderer(w, h, spp, rng_mode=gen, persist_rng=True, max_bounces=mb)
                bw = {"single": [], "batched": []}
                for win in range(5):
                    for mode in ("single", "batched"):
                        torch.cuda.synchronize()
                        t1 = time.perf_counter()
                        if mode == "batched":
                            rb.enqueue_frames(big.data_ptr(), h * w * 14, d_scene.data_ptr(), len(sph), bases, eyes, stream=stream.cuda_stream)
                        else:
                            for f in range(nb):
                                rb.enqueue(big.data_ptr() + f * h * w * 56, d_scene.data_ptr(), len(sph), basis, pt.DEFAULT_EYE, stream=stream.cuda_stream)
                        torch.cuda.synchronize()
                        bw[mode].append((time.perf_counter() - t1) / nb * 1e3)
                rb.check(wait=True)
                rb.destroy()
                frames_batch[gen_name] = {"single_launches_ms_per_frame": mmm(bw["single"]), "one_launch_per_32_frames_ms_per_frame": mmm(bw["batched"])}
            del big
        r.destroy()
        samples = w * h * spp
        out[key] = {"workload": f"{label} {w}x{h}, {spp} spp, max_bounces {mb} ({c['name']})", "frames": reps,
                    "kernel_ms": round(kms[len(kms) // 2], 4), "kernel_ms_min": round(kms[0], 4),
                    "wall_ms_per_frame": round(wall / reps * 1e3, 4),
                    "Msamples_per_s": round(samples / (wall / reps) / 1e6, 1),
                    "kernel_variant": ki["variant"], "num_vgprs": ki["num_vgprs"], "scratch_bytes": ki["scratch_bytes"],
                    "hbm_algorithmic_GBps": round(56 * w * h / (kms[len(kms) // 2] * 1e-3) / 1e9, 2)}
        if display:
            out[key]["display"] = display
            out[key]["frame_batches"] = frames_batch
        del frame, d_scene
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="cfg2")
    ap.add_argument("--engine", choices=["dist", "native"], default="dist",
                    help="dist: one process per GPU, torch.distributed over RCCL; native: one process, libptcore pt_mgpu_*")
    ap.add_argument("--rng", choices=["xorwow", "philox"], default="xorwow")
    ap.add_argument("--variant", type=int, default=None, help="kernel variant (default: the library default)")
    ap.add_argument("--spp", type=int, default=None, help="override spp (invalidates the headline config)")
    ap.add_argument("--fast", action="store_true", help="time the TOLERANCED fast mode (fast_math=1) as the main leg: a side line for profiling, never the headline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-alt-rng", action="store_true", help="skip the extra philox and fast-mode measurements")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the side record of the other BASELINE configurations")
    ap.add_argument("--cpu-rows", type=int, default=0, help="rows of the frame the CPU baseline renders (0 = size for ~8 s per build)")
    ap.add_argument("--dump", default=None, help="rank 0 saves the last gathered frame to this .npy (tests)")
    args = ap.parse_args()

    if args.gpus > 1 and args.engine == "dist" and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args))  # before anything here touches the GPU
    if os.environ.get("PT_BENCH_TEST_PIDDIR"):  # tests/test_bench_multirank_gpu.py: which processes a job consisted of ...
        open(os.path.join(os.environ["PT_BENCH_TEST_PIDDIR"], str(os.getpid())), "w").close()
    if os.environ.get("PT_BENCH_TEST_DIE_RANK") == os.environ.get("RANK", "0") and "WORLD_SIZE" in os.environ:
        os._exit(17)  # ... and a rank that dies before its first frame

    # stdout carries ONE JSON line and nothing else: libraries that print to file descriptor 1 on their own (RCCL writes a
    # version banner there when a communicator is created) are pointed at stderr until the line is printed
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    import __graft_entry__ as ge

    pt = ge.load_package()
    from cuda_pathtrace_amd import tiling

    cfg = dict(CONFIGS[args.config])
    if args.spp is not None:
        cfg["spp"] = args.spp
    WIDTH, HEIGHT, spp, MAXB = cfg["width"], cfg["height"], cfg["spp"], cfg["max_bounces"]
    headline_config = args.spp is None
    if args.fast and (args.engine == "native" or args.variant is not None):
        raise SystemExit("--fast: the fast mode has one kernel and runs in the dist engine")
    reference_scene = cfg["scene"] == "cornell"  # the alternative legs (philox, fast mode) are reported on the reference's scene

    native = args.engine == "native"
    world = 1 if native else int(os.environ.get("WORLD_SIZE", "1"))
    rank = 0 if native else int(os.environ.get("RANK", "0"))
    local_rank = 0 if native else int(os.environ.get("LOCAL_RANK", "0"))
    n_gpus = args.gpus if native else world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # PT_BENCH_SHARED_GPU=1 + PT_BENCH_BACKEND=gloo: functional test of the multi-rank path with all
    # ranks on GPU 0 (tests/test_bench_multirank_gpu.py); the driver's runs use one GPU per rank + RCCL
    shared = os.environ.get("PT_BENCH_SHARED_GPU") == "1"
    dev_index = 0 if shared else local_rank
    backend = os.environ.get("PT_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(dev_index)
    pt.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    # PT_BENCH_FORCE_DIST=1: initialise the process group even at world size 1 (exercises RCCL init,
    # barrier and all-reduce on a single-GPU box)
    use_dist = world > 1 or os.environ.get("PT_BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    rng_mode = pt.RNG_PHILOX if args.rng == "philox" else pt.RNG_XORWOW
    spheres, scene_label = scene_of(pt, cfg)
    basis = pt.camera_basis(width=WIDTH, height=HEIGHT)
    eye = pt.DEFAULT_EYE
    d_scene = torch.from_numpy(spheres.view("u1").reshape(-1).copy()).to(device)
    stream = torch.cuda.current_stream()
    total_samples = WIDTH * HEIGHT * spp * args.steps
    repaired = None  # (the native engine's band renderers are asked by pt_mgpu_render itself: a broken chain fails that call)

    if native:
        # ---- one process, libptcore's own multi-GPU entry (pt_mgpu_*) -------------------------------------
        devs = [0] * n_gpus if shared else list(range(n_gpus))
        frame = torch.empty(HEIGHT * WIDTH * 14, dtype=torch.float32, device=device)
        torch.cuda.synchronize()

        def measure(mode):
            m = pt.MultiRenderer(devs, WIDTH, HEIGHT, spp, rng_mode=mode, variant=args.variant, persist_rng=True, max_bounces=MAXB)
            for _ in range(args.warmup):
                m.render(frame.data_ptr(), d_scene.data_ptr(), len(spheres), basis, eye)
            k_ms = 0.0
            t0 = time.perf_counter()
            for _ in range(args.steps):
                m.render(frame.data_ptr(), d_scene.data_ptr(), len(spheres), basis, eye)  # synchronous: frame assembled on return
                k_ms += max(m.tile(r)["kernel_ms"] for r in range(n_gpus))
            dt = time.perf_counter() - t0
            return m, dt, k_ms / max(args.steps, 1) / 1e3

        m, elapsed, kernel_s = measure(rng_mode)
        exchange = m.backend()
        fs = m.frame_stats()  # pt_mgpu_render is synchronous: every frame is one unpipelined frame
        latency = {"frame_latency_ms": round(elapsed / args.steps * 1e3, 3), "exchange_exposed_ms": round(fs["exposed_ms"], 3),
                   "bands_per_tile": fs["bands"]} if n_gpus > 1 else None
        rb, re_ = m.tile(0)["rows"]
        # a single-device renderer of rank 0's tile only to report which kernel that tile runs
        probe = pt.Renderer(WIDTH, HEIGHT, spp, rng_mode=rng_mode, row_begin=rb, row_end=re_, variant=args.variant, persist_rng=False, max_bounces=MAXB)
        ki = probe.kernel_info(len(spheres))
        probe.destroy()
        if args.dump:
            import numpy as np

            np.save(args.dump, frame.cpu().numpy().reshape(HEIGHT, WIDTH, 14))
        alt = None
        fast = None
        others = None
        m.destroy()
        tiling_note = f"rows/{n_gpus}, one process, {exchange}" if n_gpus > 1 else "single GPU (pt_mgpu_* with one device)"
    else:
        # ---- one process per GPU, torch.distributed (backend nccl = RCCL) --------------------------------------
        # two frame/tile buffer sets: the gather of step k (RCCL stream) overlaps the render of step k+1
        fgs = [tiling.FrameGather(WIDTH, HEIGHT, device) for _ in range(2 if world > 1 else 1)]
        rb, re_ = fgs[0].rows
        step_no = [0]

        def measure(mode, fast_math=False, dump=None):
            """W untimed + K timed frames with generator `mode`: (renderer, whole-job seconds, kernel seconds), max over ranks."""
            rend = pt.Renderer(WIDTH, HEIGHT, spp, rng_mode=mode, row_begin=rb, row_end=re_, variant=None if fast_math else args.variant,
                               persist_rng=True, fast_math=fast_math, max_bounces=MAXB)
            pending = [[] for _ in fgs]

            def step(ev=None):
                slot = step_no[0] % len(fgs)
                step_no[0] += 1
                f = fgs[slot]
                f.wait_all(pending[slot])  # the gather that last used this buffer set must have finished
                if ev is not None:
                    ev[0].record(stream)
                rend.enqueue(f.tile.data_ptr(), d_scene.data_ptr(), len(spheres), basis, eye, stream=stream.cuda_stream)
                if ev is not None:
                    ev[1].record(stream)
                pending[slot] = f.gather()

            def sync():
                for slot, f in enumerate(fgs):  # every outstanding gather completes inside the timed region
                    f.wait_all(pending[slot])
                    pending[slot] = []
                if use_dist:
                    dist.barrier()
                torch.cuda.synchronize()

            for _ in range(args.warmup):
                step()
            sync()
            events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
            t0 = time.perf_counter()
            for k in range(args.steps):
                step(events[k])
            sync()
            dt = time.perf_counter() - t0
            repaired = rend.check()  # an enqueued frame whose sample-chunk chain broke is an error of this run, not of nobody
            k_ms = sum(a.elapsed_time(b) for a, b in events) / max(args.steps, 1)
            # ONE frame on its own (N > 1): render, then the gather, nothing overlapped -- the latency a caller of the synchronous
            # Renderer::Render sees ("ms/frame incl. RCCL gather"), next to the pipelined ms_per_step above
            if dump and rank == 0:  # the last TIMED frame (the latency frame below would be one frame further on)
                import numpy as np

                np.save(dump, fgs[(step_no[0] - 1) % len(fgs)].frame.cpu().numpy().reshape(HEIGHT, WIDTH, 14))
            lat = 0.0
            if world > 1:
                sync()
                t1 = time.perf_counter()
                step()
                sync()
                lat = time.perf_counter() - t1
            tmax = torch.tensor([dt, k_ms / 1e3, lat], dtype=torch.float64, device=device)
            if use_dist:
                dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            single_frame[0] = tmax[2].item()
            return rend, tmax[0].item(), tmax[1].item()

        single_frame = [0.0]
        renderer, elapsed, kernel_s = measure(rng_mode, fast_math=args.fast, dump=args.dump)
        latency = {"frame_latency_ms": round(single_frame[0] * 1e3, 3),
                   "exchange_exposed_ms": round(max(single_frame[0] - kernel_s, 0.0) * 1e3, 3)} if world > 1 else None
        ki = renderer.kernel_info(len(spheres))
        # the same measurement with the counter-based generator (north star: "a counter-based RNG in registers
        # replacing curand"); reported beside the headline, which stays on the reference's XORWOW stream
        alt = None
        if args.rng == "xorwow" and not args.no_alt_rng and reference_scene and not args.fast:
            r2, e2, k2 = measure(pt.RNG_PHILOX)
            alt = {"rng": "philox4x32-10 (counter-based, no state traffic)", "value": round(total_samples / e2 / 1e6, 2),
                   "unit": "Msamples/s", "ms_per_step": round(e2 / args.steps * 1e3, 3), "kernel_ms": round(k2 * 1e3, 3),
                   "kernel_variant": r2.kernel_info(len(spheres))["variant"]}
            r2.destroy()
        # the toleranced fast mode (FMA contraction, FP32-only intersect, hardware rsq/sin/cos; csrc/pt_fast.hip):
        # same workload, reported BESIDE the headline, which stays on the bit-exact kernel
        fast = None
        if not args.no_alt_rng and reference_scene and not args.fast:
            r3, e3, k3 = measure(rng_mode, fast_math=True)
            ki3 = r3.kernel_info(len(spheres))
            fast = {"mode": "fast_math=1: FMA contraction, FP32 cancellation-free intersect, v_rsq/v_sin/v_cos; NOT bit-exact",
                    "tolerance": "vs the ORACLE (= the exact kernel, bit for bit) at equal seeds, 256 x 256: 1 spp -- albedo identical in >= 99.8 % "
                                 "of the pixels, normals within 2e-4 in 99.9 % of them, depth within 5e-4 relative; 64 spp -- <= 3 % of the pixels "
                                 "differ by > 1e-4 in colour, median 0; image means within 4 standard errors of the MC mean "
                                 "(tests/test_fast_mode_gpu.py, table: profiles/r04/fast_vs_oracle.json)",
                    "value": round(total_samples / e3 / 1e6, 2), "unit": "Msamples/s", "ms_per_step": round(e3 / args.steps * 1e3, 3),
                    "kernel_ms": round(k3 * 1e3, 3), "rng": args.rng, "num_vgprs": ki3["num_vgprs"]}
            # counters of the fast kernel (profiles/valu_roofline.json, key cfg2_<rng>_v100): only if measured on THIS build
            if n_gpus == 1 and headline_config:
                _, vrec3, why3 = profile_records(pt, args.config, args.rng, ki3)
                if vrec3:
                    ach3 = vrec3["valu_insts_per_launch"] / k3 / 1e9
                    fast["valu_roofline"] = {"bound": "valu-issue", "achieved": round(ach3, 1), "peak": round(VALU_PEAK_GINST, 1),
                                             "unit": "G wave-instr/s", "frac": round(ach3 / VALU_PEAK_GINST, 4),
                                             "valu_insts_per_launch": vrec3["valu_insts_per_launch"],
                                             "lane_insts_per_sample": round(vrec3.get("lane_insts_per_sample", 0.0), 1),
                                             "fp32_tflops": round(vrec3["flops_fp32_per_launch"] / k3 / 1e12, 2),
                                             "source": vrec3.get("source")}
                else:
                    fast["valu_roofline"] = None
                    fast["profile_stale"] = why3
            r3.destroy()
            if args.rng == "xorwow":  # the counter-based generator in the fast mode as well
                r4, e4, k4 = measure(pt.RNG_PHILOX, fast_math=True)
                fast["philox"] = {"value": round(total_samples / e4 / 1e6, 2), "unit": "Msamples/s", "ms_per_step": round(e4 / args.steps * 1e3, 3),
                                  "kernel_ms": round(k4 * 1e3, 3), "num_vgprs": r4.kernel_info(len(spheres))["num_vgprs"]}
                r4.destroy()
        tiling_note = f"rows/{world}, one process per GPU, gather to rank 0 ({backend} grouped isend/irecv)" if world > 1 else "single GPU"
        # the other BASELINE.json configurations on this GPU (kernel time from events on the launch stream), after the headline
        others = None
        if world == 1 and args.config == "cfg2" and headline_config and not args.no_other_configs and args.variant is None:
            renderer.destroy()
            others = other_configs(pt, torch, device, stream, rng_mode)

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        tile_pixels = (re_ - rb) * WIDTH
        achieved = BYTES_PER_PIXEL * tile_pixels / kernel_s / 1e9  # GB/s of algorithmic bytes, dominant kernel
        # counters that only a profiler can measure come from profiles/*.json -- but only if that profile was taken
        # on THIS build of the library and this kernel (fingerprint + VGPR count); otherwise null, never stale numbers
        traffic, valu, stale = None, None, None
        if n_gpus == 1 and headline_config:
            hrec, vrec, stale = profile_records(pt, args.config, args.rng, ki)
            if hrec:
                traffic = hrec.get("hbm_bytes_per_launch")
            if vrec:  # instruction count per launch is a property of the code; the rate uses THIS run's kernel time
                # (a profile taken on a shorter launch of the same frame -- config 4: 16 of the 256 spp -- counts per sample)
                scale = float(WIDTH * HEIGHT * spp) / float(vrec.get("samples_per_launch") or WIDTH * HEIGHT * spp)
                if scale != 1.0:
                    vrec = dict(vrec, valu_insts_per_launch=vrec["valu_insts_per_launch"] * scale,
                                flops_fp32_per_launch=vrec["flops_fp32_per_launch"] * scale, flops_fp64_per_launch=vrec["flops_fp64_per_launch"] * scale,
                                source=f"{vrec.get('source')}, a {vrec.get('samples_per_launch') // (WIDTH * HEIGHT)}-spp launch scaled by samples")
                ach = vrec["valu_insts_per_launch"] / kernel_s / 1e9
                valu = {"bound": "valu-issue", "achieved": round(ach, 1), "peak": round(VALU_PEAK_GINST, 1), "unit": "G wave-instr/s",
                        "frac": round(ach / VALU_PEAK_GINST, 4),
                        "valu_insts_per_launch": vrec["valu_insts_per_launch"],
                        "fp32_tflops": round(vrec["flops_fp32_per_launch"] / kernel_s / 1e12, 2) if vrec["flops_fp32_per_launch"] else None,
                        "fp64_tflops": round(vrec["flops_fp64_per_launch"] / kernel_s / 1e12, 2) if vrec["flops_fp64_per_launch"] else None,
                        "source": f"{vrec.get('source')} (rocprofv3 SQ_INSTS_VALU per launch / this run's kernel time; peak = 1024 SIMDs x 2.4 GHz / 2 cycles "
                                  "per wave64 instruction, MI355X_MICROARCH.md)"}
                cw = vrec.get("class_weighted")
                if cw:  # what the kernel's own instruction mix can reach: half-rate and transcendental classes cost 4 / 8 / 16 cycles
                    valu["class_weighted"] = {"frac": cw["frac"], "costs_cycles": cw["costs_cycles"], "mix_per_sample": cw["mix_per_sample"],
                                              "clock_ghz": cw["clock_ghz"],
                                              "note": "SIMD cycles the measured mix needs at the architectural class costs / SIMD cycles the kernel "
                                                      "took, both from the profiled run (tools/issue_model.py; the 1:2:4:8 ladder was "
                                                      "measured by tools/ubench/valu_clock)"}
        out = {
            "metric": f"Msamples/s, {scene_label} {WIDTH}x{HEIGHT}x{spp}spp" + ("" if MAXB == 5 else f", {MAXB} bounces") +
                      (" [fast_math=1: toleranced side line, NOT the headline]" if args.fast else ""),
            "value": round(total_samples / elapsed / 1e6, 2),
            "unit": "Msamples/s",
            "n_gpus": n_gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32+f64",
            "data": "synthetic (reference scene include/Scene.h:26-34, default camera, fixed seed)" if cfg["scene"] == "cornell" else
                    "synthetic (pt_scene_random: seeded spheres inside the reference's box, default camera, fixed seed)",
            "config": {
                "workload": f"{scene_label} {WIDTH}x{HEIGHT}, {spp} spp, max_bounces {MAXB}, rng {args.rng} seed=pixel id ({cfg['name']})",
                "tiling": tiling_note,
                "engine": args.engine,
                "kernel_variant": ki["variant"],
            },
            "roofline": {
                "bound": "hbm",
                "achieved": round(achieved, 4),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 7),
                "traffic": traffic,
                "kernel": "pt::fast::pixel_kernel_fast" if args.fast else "pt::pixel_kernel",
                "kernel_ms": round(kernel_s * 1e3, 3),
                "note": "56 B/pixel/frame algorithmic; the kernel is VALU/latency bound (about 2.4 kFLOP per sample, "
                        "f32+f64), so the HBM fraction is <<1% by construction.  traffic = everything the counters saw: the "
                        "output, the persistent generator state (24 B in + 24 B out per pixel) and, on frames of few long "
                        "workgroups, one sample-chunk hand-over (104 B written + read per pixel: scheduling traffic, DESIGN.md 3)"
                        + ("; the many-sphere kernel (variants 13 / 14) adds three hand-overs (four sample chunks) and 48 B of register spills per lane that are "
                           "stored and reloaded every sample -- most of its traffic (DESIGN.md 4)" if ki["variant"] in (13, 14) else ""),
            },
            "valu_roofline": valu,
            "profile_stale": stale,
            "counter_based_rng": alt,
            "fast_mode": fast,
            "other_configs": others,
            "kernel_info": dict(ki, fingerprint=pt.build_fingerprint()),
            "repaired_frames": repaired,  # frames whose broken sample-chunk chain pt_renderer_render repaired in place (normally 0)
        }
        if latency:  # N > 1: one unpipelined frame (render + gather) and what the gather adds to the slowest rank's kernel
            out.update(latency)
        if n_gpus == 1 and not native and not args.no_cpu_baseline:
            oracle = ge.load_oracle()
            oracle.build()
            out["cpu_baseline"] = cpu_baseline(oracle, cfg, spheres, basis, args.cpu_rows)
        else:
            out["cpu_baseline"] = None
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)

    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
