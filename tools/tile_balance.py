#!/usr/bin/env python3
"""Kernel time of every rank's row tile at N = 2, 4, 8 (headline config, automatic variant):
the multi-GPU step time is the SLOWEST tile.  Usage: tile_balance.py"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pt = ge.load_package()
from cuda_pathtrace_amd import tiling
pt.set_device(0)
size, spp = 1024, 1024
basis = pt.camera_basis(width=size, height=size)
d_scene, ns = pt.upload_scene(pt.scene_cornell())
out = {}
for n in (1, 2, 4, 8):
    times = []
    for r in range(n):
        b, e = tiling.row_range(size, n, r)
        ren = pt.Renderer(size, size, spp, row_begin=b, row_end=e)
        d = pt.DeviceBuffer((e - b) * size * 56)
        times.append(round(min(ren.render(d.ptr, d_scene.ptr, ns, basis) for _ in range(2)), 3))
        ren.destroy(); d.free()
    out[n] = times
    print(f"N={n}: tiles {times}  max {max(times):.3f}  mean {sum(times)/n:.3f}  efficiency vs N=1: {out[1][0]/(n*max(times))*100:.1f}%", flush=True)
json.dump(out, open("gpurun_out/tile_balance.json", "w"))
