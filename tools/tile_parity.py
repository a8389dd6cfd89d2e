#!/usr/bin/env python3
"""Parity of ONE multi-GPU row tile at the headline size (what rank `r` of `n` renders), automatic
variant choice (variant 8 for small tiles), with persisted generator state over two frames.
Usage: tile_parity.py n r"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
from bench import usable_cores
pt, oracle = ge.load_package(), ge.load_oracle()
from cuda_pathtrace_amd import tiling
n, r = int(sys.argv[1]), int(sys.argv[2])
size, spp = 1024, 1024
b, e = tiling.row_range(size, n, r)
pt.set_device(0)
basis = pt.camera_basis(width=size, height=size)
ren = pt.Renderer(size, size, spp, row_begin=b, row_end=e)
d_scene, ns = pt.upload_scene(pt.scene_cornell())
d_out = pt.DeviceBuffer((e - b) * size * 56)
st = oracle.setup_random(size, size, b, e)
res = {"tile": f"rank {r} of {n}: rows [{b},{e})", "variant": ren.kernel_info(ns)["variant"], "frames": []}
for frame in range(2):
    ms = ren.render(d_out.ptr, d_scene.ptr, ns, basis)
    img = d_out.download(np.float32, (e - b, size, 14))
    t = time.perf_counter()
    ref = oracle.render(size, size, spp, spheres=pt.scene_cornell(), basis=basis, row_begin=b, row_end=e, rng_state=st, threads=usable_cores())
    neq = int((img.view(np.uint32) != ref.view(np.uint32)).sum())
    same_state = bool(np.array_equal(ren.get_rng_state(), st))
    res["frames"].append({"kernel_ms": round(ms, 3), "floats_different": neq, "state_equal": same_state, "oracle_s": round(time.perf_counter() - t, 1)})
print(json.dumps(res))
sys.exit(0 if all(f["floats_different"] == 0 and f["state_equal"] for f in res["frames"]) else 1)
