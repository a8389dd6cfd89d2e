#!/usr/bin/env python3
"""Soak for the many-sphere kernel (variants 13 and 14: grid walk, pooled tests, the sweep, per-pixel primary lists): random scenes of 72-2048 spheres -- radii
from one size to two decades apart, clustered or uniform centres, with or without the walls -- random cameras inside and outside
the cloud, both generators, 1-8 bounces, frames of complete waves; variants 13, 14 and the automatic choice against the CPU oracle,
BIT FOR BIT.  Usage: many_soak.py [n_cases=200] [first_seed=0] [large]   (large: frames of 128-256 pixels a side)"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pt, oracle = ge.load_package(), ge.load_oracle()
pt.set_device(0)
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
sizes = [128, 160, 256] if len(sys.argv) > 3 else [64, 72, 96]
bad, floats, t0, refused = [], 0, time.time(), 0
for seed in range(first, first + n_cases):
    rng = np.random.default_rng(1000003 * seed + 17)
    n = int(rng.choice([72, 96, 128, 160, 200, 300, 500, 800, 1000, 1500, 2048]))
    walls = bool(rng.integers(0, 2))
    sc = pt.scene_random(n, seed=seed, with_walls=walls)
    k0 = 7 if walls else 0
    m = len(sc) - k0
    style = int(rng.integers(0, 4))
    if style == 1:    # radii over two decades
        sc["radius"][k0:] = np.exp(rng.uniform(np.log(0.15), np.log(6.0), m)).astype(np.float32)
    elif style == 2:  # clusters: many spheres per cell in places, empty cells elsewhere
        c = rng.uniform([10, 10, 10], [90, 70, 150], (6, 3))
        sc["pos"][k0:] = (c[rng.integers(0, 6, m)] + rng.normal(0, 4.0, (m, 3))).astype(np.float32)
    elif style == 3:  # one size, touching neighbours on a jittered lattice
        g = int(np.ceil(m ** (1 / 3)))
        idx = np.stack(np.unravel_index(np.arange(m), (g, g, g)), -1).astype(np.float32)
        sc["pos"][k0:] = (10.0 + idx * (80.0 / g) + rng.uniform(-0.2, 0.2, (m, 3))).astype(np.float32)
        sc["radius"][k0:] = np.float32(40.0 / g)
    size = int(rng.choice(sizes))
    inside = bool(rng.integers(0, 2))
    eye = tuple(rng.uniform([20, 20, 20], [80, 60, 140])) if inside else tuple(rng.uniform([20, 20, 200], [80, 60, 320]))
    basis = pt.camera_basis(eye, float(rng.uniform(-130, -50)), float(rng.uniform(-25, 25)), size, size)
    mode, spp, mb = int(seed % 2), int(rng.integers(1, 5)), int(rng.integers(1, 9))
    ref = oracle.render(size, size, spp, spheres=sc, basis=basis, eye=eye, rng_mode=mode, max_bounces=mb, threads=16)
    for v in (13, 14, None):  # 14 = 13 with 1024-thread workgroups (the automatic choice above 1200 spheres on large tiles)
        img, _ = pt.render_frame(size, size, spp, spheres=sc, basis=basis, eye=eye, rng_mode=mode, max_bounces=mb, variant=v)
        neq = int((img.view(np.uint32) != ref.view(np.uint32)).sum())
        floats += img.size
        if neq:
            bad.append({"seed": seed, "variant": v, "floats_different": neq, "n": n, "style": style, "walls": walls})
    if seed % 20 == 19:
        print(f"seed {seed}: {floats} floats compared, {len(bad)} bad, {time.time() - t0:.0f} s", flush=True)
print(json.dumps({"tool": "many_soak", "cases": n_cases, "first_seed": first, "floats_compared": floats, "different": bad,
                  "fingerprint": pt.build_fingerprint(), "seconds": round(time.time() - t0, 1)}))
