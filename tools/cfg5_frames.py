#!/usr/bin/env python3
"""The interactive shape (512^2 x 4 spp x 8 bounces) as a stream of single-frame launches against pt_renderer_enqueue_frames (one
launch per 32 frames), both generators, 256 frames into separate buffers; wall ms per frame, min of 5.  Usage: cfg5_frames.py"""
import os, sys, time
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import __graft_entry__ as ge
pt = ge.load_package(); pt.set_device(0)
w = h = 512; nb = 256
basis = pt.camera_basis(width=w, height=h)
bases = np.tile(np.asarray(basis, dtype=np.float32).reshape(1, 12), (nb, 1))
eyes = np.tile(np.asarray(pt.DEFAULT_EYE, dtype=np.float32).reshape(1, 3), (nb, 1))
d_scene, n = pt.upload_scene(pt.scene_cornell())
big = pt.DeviceBuffer(nb * w * h * 56)
for rng in (pt.RNG_XORWOW, pt.RNG_PHILOX):
    r = pt.Renderer(w, h, 4, max_bounces=8, rng_mode=rng)
    res = {"single": [], "batched": []}
    for win in range(5):
        for mode in ("single", "batched"):
            pt.check(pt.lib.pt_device_synchronize())
            t = time.perf_counter()
            if mode == "batched":
                r.enqueue_frames(big.ptr, w * h * 14, d_scene.ptr, n, bases, eyes)
            else:
                for f in range(nb):
                    r.enqueue(big.ptr + f * w * h * 56, d_scene.ptr, n, basis)
            pt.check(pt.lib.pt_device_synchronize())
            res[mode].append((time.perf_counter() - t) / nb * 1e3)
    r.check(wait=True)
    print(f"rng {rng}: single launches {min(res['single']):.4f} ms per frame, one launch per 32 frames {min(res['batched']):.4f}", flush=True)
    r.destroy()
