#!/usr/bin/env python3
"""Soak of the reference-configuration kernel builds (9 spheres, 5 bounces as compile-time constants): the Cornell box
with randomly perturbed spheres, materials and cameras, both generators, variants 6, 8 and automatic, against the
CPU oracle bit for bit.  Usage: ref_config_soak.py [n_cases=3000] [first_seed=0]"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pt, oracle = ge.load_package(), ge.load_oracle()
pt.set_device(0)
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
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
    size = int(rng.choice([24, 32, 40, 48]))
    eye = tuple(rng.uniform([10, 10, 100], [90, 70, 300]))
    basis = pt.camera_basis(eye, float(rng.uniform(-130, -50)), float(rng.uniform(-25, 25)), size, size)
    mode, spp = int(seed % 2), int(rng.integers(1, 13))
    mb = 8 if seed % 3 == 0 else 5  # the two bounce caps with a compile-time build (the reference's 5, the interactive 8)
    ref = oracle.render(size, size, spp, spheres=sc, basis=basis, eye=eye, rng_mode=mode, threads=8, max_bounces=mb)
    for v in (6, 8, 9, None):
        img, _ = pt.render_frame(size, size, spp, spheres=sc, basis=basis, eye=eye, rng_mode=mode, variant=v, max_bounces=mb)
        neq = int((img.view(np.uint32) != ref.view(np.uint32)).sum())
        floats += img.size
        if neq:
            bad.append({"seed": seed, "variant": v, "floats_different": neq})
    if (seed - first) % 500 == 499:
        print(f"{seed - first + 1} cases, {floats} floats, {len(bad)} mismatching, {time.time()-t0:.0f} s", flush=True)
print(json.dumps({"cases": n_cases, "first_seed": first, "variants": [6, 8, 9, "auto"], "max_bounces": "5, every third case 8", "floats_compared": floats, "mismatches": bad}))
sys.exit(1 if bad else 0)
