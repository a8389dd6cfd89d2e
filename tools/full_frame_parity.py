#!/usr/bin/env python3
"""Whole-frame parity at the headline size: renders BASELINE.json config 2 (1024x1024x1024 spp,
1.07e9 samples) with the HIP kernel and with the CPU oracle (all usable cores, ~30 s on the GPU
box) and compares all 14.7 M floats bit for bit.  Too slow for the unit tests; run once per
round, result recorded under profiles/.  Usage: full_frame_parity.py [rng] [variant]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
from bench import usable_cores

pt = ge.load_package()
oracle = ge.load_oracle()
rng = int(sys.argv[1]) if len(sys.argv) > 1 else 0
variant = int(sys.argv[2]) if len(sys.argv) > 2 else None
size, spp = 1024, 1024
pt.set_device(0)
basis = pt.camera_basis(width=size, height=size)
opts = dict(rng_mode=rng)
if variant is not None:
    opts["variant"] = variant
img, ms = pt.render_frame(size, size, spp, basis=basis, **opts)
t = time.perf_counter()
ref = oracle.render(size, size, spp, spheres=pt.scene_cornell(), basis=basis, rng_mode=rng, threads=usable_cores())
dt = time.perf_counter() - t
neq = img.view(np.uint32) != ref.view(np.uint32)
res = {
    "config": f"{size}x{size}x{spp}spp Cornell, rng {'philox' if rng else 'xorwow'}, variant {variant if variant is not None else 'default'}",
    "gpu_kernel_ms": round(ms, 3), "oracle_seconds": round(dt, 1), "oracle_threads": usable_cores(),
    "floats_compared": int(neq.size), "floats_different": int(neq.sum()), "pixels_different": int(neq.any(axis=2).sum()),
    "max_abs_diff": float(np.nanmax(np.abs(img - ref))), "channel_means": [float(x) for x in img.reshape(-1, 14).mean(0, dtype=np.float64)],
}
print(json.dumps(res))
sys.exit(0 if res["floats_different"] == 0 else 1)
