#!/usr/bin/env python3
"""One-off soak: N random scenes/cameras/generators (same generator as tests/test_parity_gpu.py's fuzz
test), variants 0, 6, 8, 10, 13, 14 and the automatic choice against the CPU oracle, bit for bit.
Usage: fuzz_soak.py [n_seeds] [first_seed]"""
import json, os, sys, time
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import __graft_entry__ as ge
from test_parity_gpu import _random_scene
pt, oracle = ge.load_package(), ge.load_oracle()
pt.set_device(0)
n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
first = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
bad, floats, t0 = [], 0, time.time()
for seed in range(first, first + n_seeds):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(1, 40)) if seed % 5 else int(rng.integers(65, 200))
    scene = _random_scene(rng, n)
    size = int(rng.choice([24, 32, 40, 48]))
    eye = tuple(rng.uniform([20, 20, 100], [80, 60, 300]))
    basis = pt.camera_basis(eye, float(rng.uniform(-120, -60)), float(rng.uniform(-20, 20)), size, size)
    mode, spp, mb = int(seed % 2), int(rng.integers(1, 12)), int(rng.integers(1, 9))
    ref = oracle.render(size, size, spp, spheres=scene, basis=basis, eye=eye, rng_mode=mode, max_bounces=mb, threads=8)
    for v in (0, 6, 8, 10, 13, 14, None):
        img, _ = pt.render_frame(size, size, spp, spheres=scene, basis=basis, eye=eye, rng_mode=mode, max_bounces=mb, variant=v)
        neq = int((img.view(np.uint32) != ref.view(np.uint32)).sum())
        floats += img.size
        if neq:
            bad.append({"seed": seed, "variant": v, "floats_different": neq})
    if (seed - first) % 100 == 99:
        print(f"{seed - first + 1} seeds, {floats} floats compared, {len(bad)} mismatching (seed, variant) pairs, {time.time()-t0:.0f} s", flush=True)
res = {"seeds": n_seeds, "first_seed": first, "variants": [0, 6, 8, 10, 13, 14, "auto"], "floats_compared": floats, "mismatches": bad}
print(json.dumps(res))
sys.exit(1 if bad else 0)
