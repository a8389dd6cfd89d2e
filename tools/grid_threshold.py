#!/usr/bin/env python3
"""Where should scenes switch from the brute-force regeneration kernel (variant 10) to the grid kernel (variant 13)?  1024^2 x 32 spp,
n random spheres with and without the walls: the automatic choice, variant 10, variant 13 (PT_GRID_MIN_SPHERES_LARGE_TILE, csrc/pt_kernel.h).
Usage: grid_threshold.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pt = ge.load_package(); pt.set_device(0)
spp = 32
basis = pt.camera_basis(width=1024, height=1024)
d_out = pt.DeviceBuffer(1024 * 1024 * 56)
for n in (40, 56, 64, 72, 80, 100, 120, 140, 160, 200):
    for walls in (True, False):
        scene = pt.scene_random(n, seed=1, with_walls=walls)
        d_scene, ns = pt.upload_scene(scene)
        res = []
        for var in (None, 10, 13):
            r = pt.Renderer(1024, 1024, spp, variant=var)
            ms = min(r.render(d_out.ptr, d_scene.ptr, ns, basis) for _ in range(3))
            res.append(f"{'auto=' + str(r.kernel_info(ns)['variant']) if var is None else 'v' + str(var)} {ms:6.3f}")
            r.destroy()
        print(f"n {n:4d} {'closed' if walls else 'open  '}: " + " | ".join(res), flush=True)
