#!/usr/bin/env python3
"""Scenes at and above the size the 512-thread grid kernel's LDS image was made for: n random spheres (with and without the walls),
1024^2 x 32 spp, variant 13 (two 512-thread workgroups per CU share its LDS: the cell table gets what 80 KB leave) against variant 14
(one 1024-thread workgroup per CU: one image, the table gets the other half) and the automatic choice.  Usage: large_scenes.py [spp=32]"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import __graft_entry__ as ge
pt = ge.load_package(); pt.set_device(0)
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 32
basis = pt.camera_basis(width=1024, height=1024)
d_out = pt.DeviceBuffer(1024 * 1024 * 56)
for n in (1000, 1200, 1500, 2000, 2048):
    for walls in (True, False):
        d_scene, ns = pt.upload_scene(pt.scene_random(n, seed=1, with_walls=walls))
        res = []
        for v in (13, 14, None):
            r = pt.Renderer(1024, 1024, spp, variant=v)
            ms = min(r.render(d_out.ptr, d_scene.ptr, ns, basis) for _ in range(2))
            res.append(f"{'auto=' + str(r.kernel_info(ns)['variant']) if v is None else v}: {ms:7.2f}")
            r.destroy()
        print(f"{n:5d} spheres {'closed' if walls else 'open  '} spp {spp}  ms  " + "   ".join(res), flush=True)
        d_scene.free()
