#!/usr/bin/env python3
"""Soak of the pixel-footprint sphere exclusion (pt_footprint.h; REF kernel = 9 spheres, 5 bounces, variant 6, spp >= 8): the
Cornell box with randomly perturbed spheres (moved / resized small spheres and light, shifted walls, spheres poking through
walls, the eye close to a sphere) and random cameras, rendered at resolutions where the footprint test is active
(256..768 pixels) and compared with the CPU oracle bit for bit on whole frames.
Usage: footprint_soak.py [n_cases=150] [first_seed=0] [alt build name]"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 3:  # an alternative build (mutation testing): cuda-pathtrace_amd/alt/<name>/libptcore.so
    os.environ["PT_LIB_OVERRIDE"] = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cuda-pathtrace_amd", "alt", sys.argv[3], "libptcore.so")
import __graft_entry__ as ge
pt, oracle = ge.load_package(), ge.load_oracle()
pt.set_device(0)
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad, floats, t0 = [], 0, time.time()
base = pt.scene_cornell()
for seed in range(first, first + n_cases):
    rng = np.random.default_rng(70000 + seed)
    sc = base.copy()
    k = seed % 10
    if k >= 1:  # move and resize the two small spheres and the light
        sc["pos"][6:] += rng.normal(0, 8.0, size=(3, 3)).astype(np.float32)
        sc["radius"][6:8] *= rng.uniform(0.3, 2.5, 2).astype(np.float32)
    if k >= 2:  # shift walls by a few units (junctions move, the light cap changes)
        sc["pos"][:6] += rng.normal(0, 1.5, size=(6, 3)).astype(np.float32)
    if k == 3:  # a small sphere poking through a wall / the floor
        sc["pos"][6] = (rng.uniform(0, 8), rng.uniform(0, 20), rng.uniform(20, 150))
    if k == 4:  # the light far bigger / lower
        sc["pos"][8, 1] -= rng.uniform(0, 3)
    if k == 6:  # nested and duplicated spheres (ties, first-index rule), a sphere the eye is inside of but small
        sc["pos"][7] = sc["pos"][6]
        sc["radius"][7] = sc["radius"][6] * (1.0 if seed % 20 == 6 else rng.uniform(0.5, 1.2))
    if k == 7:  # a mid-sized sphere around the eye region: "inside" type that is NOT a wall, often grazing
        sc["radius"][6] = rng.uniform(60, 400)
    scale = 1.0
    if k == 8:  # the whole scene scaled (absolute thresholds would show)
        scale = float(rng.choice([0.01, 100.0]))
        sc["pos"] *= np.float32(scale)
        sc["radius"] *= np.float32(scale)
    size = int(rng.choice([256, 320, 500, 512, 768, 1024]))  # 320/500: the non-power-of-two divide path
    if k == 5:  # eye close to a small sphere (large angular size, grazing views, c near 0)
        eye = tuple((sc["pos"][6] + rng.normal(0, 1, 3) * (sc["radius"][6] * rng.choice([1.001, 1.02, 1.3]) + rng.choice([0.0, 5.0]))).astype(float))
    elif k == 9:  # eye hugging a wall or in a corner
        eye = tuple(rng.choice([[1.2, 40, 100], [98.9, 5, 20], [50, 81.3, 150], [2, 1, 2], [50, 40, 598]]) + rng.normal(0, 0.05, 3))
    else:
        eye = tuple(np.array(rng.uniform([10, 10, 120], [90, 70, 320])) * scale)
    basis = pt.camera_basis(eye, float(rng.uniform(-180, 180) if k == 9 else rng.uniform(-130, -50)), float(rng.uniform(-40, 40)), size, size)
    mode, spp = int(seed % 2), int(rng.choice([8, 9, 16]))
    rows = (0, size) if size <= 512 else tuple(sorted(rng.integers(0, size - 64, 1)))[0:1] * 1
    rb = 0 if size <= 512 else int(rng.integers(0, size - 128))
    re_ = size if size <= 512 else rb + 128  # large frames: a 128-row tile
    ref = oracle.render(size, size, spp, spheres=sc, basis=basis, eye=eye, rng_mode=mode, row_begin=rb, row_end=re_)
    for v in (6, 8):
        img, _ = pt.render_frame(size, size, spp, spheres=sc, basis=basis, eye=eye, rng_mode=mode, variant=v, row_begin=rb, row_end=re_)
        neq = int((img.view(np.uint32) != ref.view(np.uint32)).sum())
        floats += img.size
        if neq:
            bad.append({"seed": seed, "variant": v, "floats_different": neq, "size": size, "kind": int(k)})
    if (seed - first) % 25 == 24:
        print(f"{seed - first + 1} cases, {floats} floats, {len(bad)} mismatching, {time.time()-t0:.0f} s", flush=True)
res = {"cases": n_cases, "first_seed": first, "floats_compared": floats, "mismatches": bad}
print(json.dumps(res))
sys.exit(1 if bad else 0)
