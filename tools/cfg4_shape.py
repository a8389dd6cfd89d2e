#!/usr/bin/env python3
"""Config 4 (1000 spheres) on frames of equal sample count but different shape: does the grid kernel lose time to few, long
workgroups like the headline kernel did (tools/shape_sweep.py)?  With the progress priorities: no (closed 132.8 / 132.3 / 127.9 ms
at 1024^2x256 / 2048^2x64 / 4096^2x16), so sample chunking is not extended to it."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pt = ge.load_package(); pt.set_device(0)
for walls in (True, False):
    scene = pt.scene_random(1000, seed=1, with_walls=walls)
    d_scene, n = pt.upload_scene(scene)
    for size, spp in ((512, 1024), (1024, 256), (2048, 64), (4096, 16)):
        basis = pt.camera_basis(width=size, height=size)
        r = pt.Renderer(size, size, spp)
        d_out = pt.DeviceBuffer(size * size * 56)
        ms = min(r.render(d_out.ptr, d_scene.ptr, n, basis) for _ in range(2))
        print(f"{'closed' if walls else 'open'} {size}^2x{spp}: {ms:.2f} ms v{r.kernel_info(n)['variant']}", flush=True)
        r.destroy(); d_out.free()
