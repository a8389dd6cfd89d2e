#!/usr/bin/env python3
"""Kernel times for the BASELINE.json configs on one GPU, including the row tiles a rank
renders at N = 2/4/8 GPUs (predicts strong-scaling of the kernel part).  Usage:
  python tools/config_times.py [variant]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge

pt = ge.load_package()
pt.set_device(0)
variant = int(sys.argv[1]) if len(sys.argv) > 1 else None  # None = the library's automatic choice
out = {"variant": variant, "device": pt.device_info()}


def run(name, size, spp, spheres, reps=3, rows=None, **kw):
    basis = pt.camera_basis(width=size, height=size)
    rb, re_ = rows if rows else (0, size)
    r = pt.Renderer(size, size, spp, variant=variant, row_begin=rb, row_end=re_, **kw)
    d_scene, n = pt.upload_scene(spheres)
    d_out = pt.DeviceBuffer((re_ - rb) * size * 14 * 4)
    ms = [r.render(d_out.ptr, d_scene.ptr, n, basis) for _ in range(reps)]
    used = r.kernel_info(n)["variant"]
    r.destroy()
    samples = (re_ - rb) * size * spp
    res = {"variant": used, "ms_min": round(min(ms), 3), "ms_all": [round(m, 3) for m in ms], "Msamples_per_s": round(samples / min(ms) / 1e3, 1)}
    out[name] = res
    print(name, res, flush=True)


cornell = pt.scene_cornell()
run("cfg2_1024x1024x1024spp_full", 1024, 1024, cornell)
for n in (2, 4, 8):
    run(f"cfg2_tile_1_of_{n}_rows", 1024, 1024, cornell, rows=(0, 1024 // n))
run("cfg3_tile_4096x512rows_x64spp (1 of 8)", 4096, 64, cornell, rows=(0, 512))
run("cfg4_1000spheres_closed_1024x1024x256spp", 1024, 256, pt.scene_random(1000, seed=1, with_walls=True), reps=2)
run("cfg4_1000spheres_open_1024x1024x256spp", 1024, 256, pt.scene_random(1000, seed=1, with_walls=False), reps=2)
run("cfg5_512x512x4spp_8bounces_per_frame", 512, 4, cornell, reps=20, max_bounces=8)
run("cfg5_philox", 512, 4, cornell, reps=20, max_bounces=8, rng_mode=pt.RNG_PHILOX)
run("cfg1_256x256x4spp", 256, 4, cornell, reps=10)


def frame_stream(name, size, spp, frames=200, **kw):
    """cfg5's shape: frames enqueued back to back on one stream, one sync at the end -- is the loop launch-bound?"""
    import time
    basis = pt.camera_basis(width=size, height=size)
    r = pt.Renderer(size, size, spp, variant=variant, **kw)
    d_scene, n = pt.upload_scene(cornell)
    d_out = pt.DeviceBuffer(size * size * 14 * 4)
    kernel_ms = min(r.render(d_out.ptr, d_scene.ptr, n, basis) for _ in range(5))
    pt.check(pt.lib.pt_device_synchronize())
    t = time.perf_counter()
    for _ in range(frames):
        r.enqueue(d_out.ptr, d_scene.ptr, n, basis)
    t_enq = time.perf_counter() - t
    pt.check(pt.lib.pt_device_synchronize())
    wall = time.perf_counter() - t
    r.destroy()
    res = {"frames": frames, "kernel_ms": round(kernel_ms, 4), "wall_ms_per_frame": round(wall / frames * 1e3, 4),
           "host_enqueue_us_per_frame": round(t_enq / frames * 1e6, 2)}
    out[name] = res
    print(name, res, flush=True)


frame_stream("cfg5_stream_200_frames_512x512x4spp_8bounces", 512, 4, max_bounces=8)
frame_stream("cfg5_stream_200_frames_philox", 512, 4, max_bounces=8, rng_mode=pt.RNG_PHILOX)
# PCIe-inclusive single-frame mode (main.cu:187-188): D2H copy of the 58.7 MB buffer into pageable host memory
import time
buf = pt.DeviceBuffer(1024 * 1024 * 56)
host = np.empty(1024 * 1024 * 14, dtype=np.float32)
ts = []
for _ in range(5):
    t = time.perf_counter()
    pt.check(pt.lib.pt_memcpy_d2h(host.ctypes.data, buf.ptr, host.nbytes))
    ts.append((time.perf_counter() - t) * 1e3)
out["d2h_58.7MB_ms"] = [round(x, 3) for x in ts]
print("d2h 58.7 MB ms", out["d2h_58.7MB_ms"], flush=True)
json.dump(out, open(os.path.join("gpurun_out", f"config_times_v{variant if variant is not None else 'auto'}.json"), "w"), indent=1)
