#!/usr/bin/env python3
"""Variant 12 against variant 11 (bit for bit) on the config-4 scenes at 1024^2 and a few odd shapes.  Usage: v12_check.py [spp=4]"""
import os, sys
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import __graft_entry__ as ge
pt = ge.load_lab(); pt.set_device(0)  # variant 12 is an experiment: lab library only
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 4
bad = 0
for (w, h, nsph, walls, rng_mode, mb) in [(1024, 1024, 1000, True, 0, 5), (1024, 1024, 1000, False, 0, 5), (1024, 1024, 1000, True, 1, 5),
                                          (333, 77, 500, True, 0, 3), (64, 64, 2000, False, 1, 8), (200, 200, 300, True, 0, 1)]:
    scene = pt.scene_random(nsph, seed=3, with_walls=walls)
    basis = pt.camera_basis(width=w, height=h)
    d_scene, n = pt.upload_scene(scene)
    frames = []
    for v in (11, 12):
        r = pt.Renderer(w, h, spp, variant=v, rng_mode=rng_mode, max_bounces=mb)
        d_out = pt.DeviceBuffer(w * h * 56)
        outs = []
        for f in range(2):  # two frames: the persisted generator state too
            ms = r.render(d_out.ptr, d_scene.ptr, n, basis)
            outs.append(d_out.download(np.float32, (h, w, 14)).copy())
        frames.append((outs, ms))
        r.destroy()
    same = all(np.array_equal(a.view(np.uint32), b.view(np.uint32)) for a, b in zip(frames[0][0], frames[1][0]))
    bad += 0 if same else 1
    print(f"{w}x{h} n={n} walls={walls} rng={rng_mode} mb={mb}: v11 {frames[0][1]:.3f} ms, v12 {frames[1][1]:.3f} ms, {'identical' if same else 'DIFFERENT'}", flush=True)
sys.exit(1 if bad else 0)
