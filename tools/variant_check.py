#!/usr/bin/env python3
"""Run every kernel variant on the GPU at a given size, check that all variants give
bit-identical frames (variant 0 is the literal transcription that the oracle tests pin), and
print kernel times.  Usage: python tools/variant_check.py [size] [spp] [reps] [n_spheres]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge

pt = ge.load_lab()  # every variant, including the experimental ones
size = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 64
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
nsph = int(sys.argv[4]) if len(sys.argv) > 4 else 0
max_bounces = int(sys.argv[5]) if len(sys.argv) > 5 else 5
pt.set_device(0)
spheres = pt.scene_random(nsph, seed=1, with_walls=True) if nsph else pt.scene_cornell()
basis = pt.camera_basis(width=size, height=size)
d_scene, n = pt.upload_scene(spheres)
d_out = pt.DeviceBuffer(size * size * 14 * 4)
nvar = 0
while True:
    try:
        pt.Renderer(8, 8, 1, variant=nvar).destroy()
        nvar += 1
    except pt.PtError:
        break
ok = True
for rng in (pt.RNG_XORWOW, pt.RNG_PHILOX):
    ref = None
    for v in range(nvar):
        r = pt.Renderer(size, size, spp, rng_mode=rng, variant=v, persist_rng=False, max_bounces=max_bounces)
        ms = [r.render(d_out.ptr, d_scene.ptr, n, basis) for _ in range(reps)]
        img = d_out.download(np.float32, (size, size, 14))
        ki = r.kernel_info(n)
        r.destroy()
        if ref is None:
            ref = img
            same = "ref"
        else:
            neq = img.view(np.uint32) != ref.view(np.uint32)
            same = "bit-exact" if not neq.any() else f"MISMATCH {neq.sum()} floats in {neq.any(axis=2).sum()} pixels, max|d|={np.nanmax(np.abs(img-ref)):.3g}"
            ok &= not neq.any()
        msamp = size * size * spp / (min(ms) * 1e-3) / 1e6
        print(f"rng={'xorwow' if rng == 0 else 'philox'} variant={v} vgpr={ki['num_vgprs']} scratch={ki['scratch_bytes']} "
              f"min {min(ms):.3f} ms  {msamp:.0f} Msamples/s  {same}", flush=True)
sys.exit(0 if ok else 1)
