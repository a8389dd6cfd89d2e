#!/usr/bin/env python3
"""Soak of SAMPLE CHUNKING (reference-configuration variant 6 at 512+ samples per pixel: a pixel's samples go through 8
workgroups of one launch): the perturbed Cornell boxes of ref_config_soak.py at 512..900 spp, odd image sizes (ragged last
workgroup), row tiles, both generators, variant 6, against the CPU oracle bit for bit.
Usage: chunk_soak.py [n_cases=300] [first_seed=0]"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pt, oracle = ge.load_package(), ge.load_oracle()
pt.set_device(0)
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad, floats, t0 = [], 0, time.time()
base = pt.scene_cornell()
for seed in range(first, first + n_cases):
    rng = np.random.default_rng(seed)
    sc = base.copy()
    k = seed % 4
    if k >= 1:  # move and resize the three small spheres
        sc["pos"][6:] += rng.normal(0, 6.0, size=(3, 3)).astype(np.float32)
        sc["radius"][6:8] *= np.float32(rng.uniform(0.3, 1.6))
    if k >= 2:  # random materials, a second light
        sc["color"] = rng.uniform(0.0, 1.0, size=(9, 3)).astype(np.float32)
        sc["emission"][int(rng.integers(0, 9))] = rng.uniform(0, 6, size=3).astype(np.float32)
    if k == 3:  # open one wall (rays escape), shuffle the order
        sc["radius"][int(rng.integers(0, 6))] = np.float32(rng.uniform(1.0, 30.0))
        sc = sc[rng.permutation(9)]
    size = int(rng.choice([17, 24, 31, 40]))
    eye = tuple(rng.uniform([10, 10, 100], [90, 70, 300]))
    basis = pt.camera_basis(eye, float(rng.uniform(-130, -50)), float(rng.uniform(-25, 25)), size, size)
    mode, spp = int(seed % 2), int(rng.integers(512, 901))
    r0 = int(rng.integers(0, size // 2))
    r1 = int(rng.integers(r0 + 1, size + 1))
    ref = oracle.render(size, size, spp, spheres=sc, basis=basis, eye=eye, rng_mode=mode, threads=8, row_begin=r0, row_end=r1)
    for v in (6, 8, 9):
        img, _ = pt.render_frame(size, size, spp, spheres=sc, basis=basis, eye=eye, rng_mode=mode, variant=v, row_begin=r0, row_end=r1)
        neq = int((img.view(np.uint32) != ref.view(np.uint32)).sum())
        floats += img.size
        if neq:
            bad.append({"seed": seed, "variant": v, "floats_different": neq})
    if (seed - first) % 50 == 49:
        print(f"{seed - first + 1} cases, {floats} floats, {len(bad)} mismatching, {time.time()-t0:.0f} s", flush=True)
print(json.dumps({"cases": n_cases, "first_seed": first, "variants": [6, 8, 9], "spp": "512..900", "floats_compared": floats, "mismatches": bad}))
sys.exit(1 if bad else 0)
