#!/usr/bin/env python3
"""Short frames (no sample chunking): variants 6, 8, 9 and the automatic choice by frame size and samples per pixel.
Usage: short_frames.py [rng=1 (philox) | 0 (xorwow)]"""
import os, sys
sys.path.insert(0, '/root/repo')
import __graft_entry__ as ge
pt = ge.load_package(); pt.set_device(0)
rng = int(sys.argv[1]) if len(sys.argv) > 1 else 1
name = 'philox' if rng else 'xorwow'
d_scene, n = pt.upload_scene(pt.scene_cornell())
for size in (256, 320, 384, 448, 512, 576, 640):
    b = pt.camera_basis(width=size, height=size); d = pt.DeviceBuffer(size*size*56)
    for spp, mb in ((4, 8), (8, 8), (16, 5), (64, 5), (256, 5)):
        res = {}
        for v in (6, 8, 9, None):
            r = pt.Renderer(size, size, spp, rng_mode=rng, variant=v, max_bounces=mb)
            ms = sorted(r.render(d.ptr, d_scene.ptr, n, b) for _ in range(12))[0]
            res[f"auto={r.kernel_info(n)['variant']}" if v is None else f"v{v}"] = round(ms, 4); r.destroy()
        print(f"{name} {size}^2 (w={size*size/65536:.2f}) x {spp} spp x {mb} bounces: {res}", flush=True)
