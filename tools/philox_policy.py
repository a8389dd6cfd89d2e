#!/usr/bin/env python3
"""Philox: variants 6, 8, 9 by tile size of the headline frame (1024 spp, 5 bounces) and on the config-5 shape (512^2 x 4 spp x 8
bounces), kernel ms, and what the automatic policy picks.  Usage: philox_policy.py [rows...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pt = ge.load_package(); pt.set_device(0)
rows_list = [int(x) for x in sys.argv[1:]] or [64, 128, 192, 256, 320, 384, 512, 640, 768, 1024]
basis = pt.camera_basis(width=1024, height=1024)
d_scene, n = pt.upload_scene(pt.scene_cornell())
d_out = pt.DeviceBuffer(1024 * 1024 * 56)
print("build", pt.build_fingerprint(), flush=True)
for rows in rows_list:
    res = {}
    for v in (6, 8, 9, None):
        r = pt.Renderer(1024, 1024, 1024, rng_mode=1, variant=v, row_begin=0, row_end=rows)
        ms = min(r.render(d_out.ptr, d_scene.ptr, n, basis) for _ in range(3))
        res[f"auto={r.kernel_info(n)['variant']}" if v is None else f"v{v}"] = round(ms, 3)
        r.destroy()
    print(f"philox 1024 spp rows {rows:5d} ({rows / 64.0:5.2f} one-lane waves per SIMD): {res}", flush=True)
b5 = pt.camera_basis(width=512, height=512)
d5 = pt.DeviceBuffer(512 * 512 * 56)
for spp in (4, 8, 16, 64):
    res = {}
    for v in (6, 8, 9, None):
        r = pt.Renderer(512, 512, spp, rng_mode=1, variant=v, max_bounces=8)
        ms = sorted(r.render(d5.ptr, d_scene.ptr, n, b5) for _ in range(20))[0]
        res[f"auto={r.kernel_info(n)['variant']}" if v is None else f"v{v}"] = round(ms, 4)
        r.destroy()
    print(f"philox 512^2 x {spp} spp x 8 bounces: {res}", flush=True)
