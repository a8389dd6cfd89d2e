#!/usr/bin/env python3
"""Scenes the exactness arguments talk about but the random soaks never produce: spheres with NaN / infinite / huge / zero /
negative / denormal-size data, ray origins exactly on surfaces with tangent directions (c = 0, h = 0), duplicated spheres.
The Cornell box with one to three of its nine spheres replaced (the reference-configuration builds: variants 6, 8, 9, auto)
12-sphere scenes (generic builds: 0, 6, 8, 10) and 170-400-sphere scenes (10, 13), both generators, against the CPU oracle BIT FOR BIT, NaN patterns
included (np.uint32 views).  Usage: degenerate_soak.py [n_cases=400] [first_seed=0]"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pt, oracle = ge.load_package(), ge.load_oracle()
pt.set_device(0)
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 400
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
F = np.float32
ODD = [np.nan, np.inf, -np.inf, 0.0, -0.0, 1e30, -1e30, 3e38, 1e-30, 1e-42, 1e19, -1e19, 1e-20]
bad, floats, t0 = [], 0, time.time()
base = pt.scene_cornell()
for seed in range(first, first + n_cases):
    rng = np.random.default_rng(seed)
    sc = base.copy()
    if seed % 7 == 3:  # a many-sphere scene: the grid kernels (11, 13) and the scalar-load loop (10)
        sc = pt.scene_random(int(rng.integers(170, 400)), seed=seed, with_walls=bool(seed % 2))
    elif seed % 2 == 1:  # 12 spheres: the generic builds
        extra = base[[6, 7, 8]].copy()
        extra["pos"] += rng.normal(0, 8.0, size=(3, 3)).astype(F)
        sc = np.concatenate([sc, extra])
    n = len(sc)
    for _ in range(int(rng.integers(1, 4))):
        i = int(rng.integers(0, n))
        kind = int(rng.integers(0, 6))
        if kind == 0:
            sc["pos"][i, int(rng.integers(0, 3))] = F(rng.choice(ODD))
        elif kind == 1:
            sc["radius"][i] = F(rng.choice(ODD))
        elif kind == 2:  # a duplicate of another sphere (ties)
            sc[i] = sc[int(rng.integers(0, n))]
        elif kind == 3:  # tiny sphere
            sc["radius"][i] = F(10.0 ** rng.uniform(-25, -3))
        elif kind == 4:  # gigantic sphere through the scene
            sc["radius"][i] = F(10.0 ** rng.uniform(6, 18))
        else:
            sc["emission"][i] = F(rng.choice(ODD))
    size = int(rng.choice([16, 24, 32]))
    eye = np.array(rng.uniform([10, 10, 100], [90, 70, 300]), dtype=np.float64)
    if seed % 5 == 0:  # the eye EXACTLY on a sphere's surface, looking along a tangent (c = 0, h = 0 for the centre pixel's axis)
        j = int(rng.integers(6, 8)) if n < 100 else int(rng.integers(0, n))
        eye = (sc["pos"][j].astype(np.float64) + np.array([float(sc["radius"][j]), 0.0, 0.0]))
    eye = tuple(float(F(x)) for x in eye)
    basis = pt.camera_basis(eye, float(rng.uniform(-130, -50)), float(rng.uniform(-25, 25)), size, size)
    mode, spp = int(seed % 2), int(rng.integers(1, 9))
    mb = 8 if seed % 3 == 0 else 5
    with np.errstate(all="ignore"):
        ref = oracle.render(size, size, spp, spheres=sc, basis=basis, eye=eye, rng_mode=mode, threads=8, max_bounces=mb)
    variants = (6, 8, 9, None) if n == 9 else ((0, 6, 8, 10, None) if n < 100 else (10, 13, 14, None))
    for v in variants:
        img, _ = pt.render_frame(size, size, spp, spheres=sc, basis=basis, eye=eye, rng_mode=mode, variant=v, max_bounces=mb)
        a, b = img.view(np.uint32), ref.view(np.uint32)
        nan_both = np.isnan(img) & np.isnan(ref)  # NaN payloads and signs are not part of the contract
        neq = int(((a != b) & ~nan_both).sum())
        floats += img.size
        if neq:
            bad.append({"seed": seed, "variant": v, "floats_different": neq, "n": n})
    if (seed - first) % 100 == 99:
        print(f"{seed - first + 1} cases, {floats} floats, {len(bad)} mismatching, {time.time()-t0:.0f} s", flush=True)
print(json.dumps({"cases": n_cases, "first_seed": first, "floats_compared": floats, "mismatches": bad[:40], "n_mismatching": len(bad)}))
sys.exit(1 if bad else 0)
