#!/usr/bin/env python3
"""Soak of the per-pixel primary-ray lists of the many-sphere kernel (csrc/pt_primlist.h; variant 13, spp >= 4): random scenes of
72-1200 spheres with and without walls, 4-9 spp, frames whose width is or is not a multiple of 64 (a wave that straddles rows gets
no lists), and cameras chosen to sit where the cone test decides: far outside the cloud, inside it, a hair outside a sphere's
surface, looking along rows of spheres (lists overflow), with spheres behind the eye.  Variant 13 against the CPU oracle, BIT FOR
BIT.  Usage: primlist_soak.py [n_cases=200] [first_seed=0]      (PT_LIB_ALT=<name>: cuda-pathtrace_amd/alt/<name>/libptcore.so --
the deliberately unsound builds -DPT_PRIMLIST_MUTANT=1|2 must FAIL this soak)"""
import json, os, sys, time
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
if os.environ.get("PT_LIB_ALT"):
    os.environ["PT_LIB_OVERRIDE"] = os.path.join(root, "cuda-pathtrace_amd", "alt", os.environ["PT_LIB_ALT"], "libptcore.so")
import __graft_entry__ as ge
pt, oracle = ge.load_package(), ge.load_oracle()
pt.set_device(0)
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad, floats, t0 = [], 0, time.time()
for seed in range(first, first + n_cases):
    rng = np.random.default_rng(7000003 * seed + 5)
    n = int(rng.choice([72, 100, 160, 300, 600, 1000, 1200]))
    walls = bool(rng.integers(0, 2))
    sc = pt.scene_random(n, seed=seed, with_walls=walls)
    k0 = 7 if walls else 0
    m = len(sc) - k0
    style = int(rng.integers(0, 5))
    if style == 1:    # radii over two decades
        sc["radius"][k0:] = np.exp(rng.uniform(np.log(0.15), np.log(6.0), m)).astype(np.float32)
    elif style == 2:  # rows of spheres along z: many in one pixel's cone when the camera looks down the rows
        cols = rng.uniform([10, 10], [90, 70], (12, 2))
        pick = rng.integers(0, 12, m)
        sc["pos"][k0:, 0:2] = (cols[pick] + rng.normal(0, 0.3, (m, 2))).astype(np.float32)
        sc["pos"][k0:, 2] = rng.uniform(0, 170, m).astype(np.float32)
        sc["radius"][k0:] = rng.uniform(0.8, 2.0, m).astype(np.float32)
    elif style == 3:  # one size on a jittered lattice
        g = int(np.ceil(m ** (1 / 3)))
        idx = np.stack(np.unravel_index(np.arange(m), (g, g, g)), -1).astype(np.float32)
        sc["pos"][k0:] = (10.0 + idx * (80.0 / g) + rng.uniform(-0.2, 0.2, (m, 3))).astype(np.float32)
        sc["radius"][k0:] = np.float32(30.0 / g)
    cam = int(rng.integers(0, 5))
    if cam == 0:      # the reference's kind of view: outside, looking in
        eye = tuple(rng.uniform([20, 20, 200], [80, 60, 320]))
    elif cam == 1:    # inside the cloud
        eye = tuple(rng.uniform([20, 20, 20], [80, 60, 140]))
    elif cam == 2:    # a hair outside a sphere's surface (1e-4 .. 0.5 radii)
        j = k0 + int(rng.integers(0, m))
        v = rng.normal(size=3); v /= np.linalg.norm(v)
        eye = tuple(np.asarray(sc["pos"][j], dtype=np.float64) + v * float(sc["radius"][j]) * (1.0 + 10.0 ** rng.uniform(-4, -0.3)))
    elif cam == 3:    # far away: small spheres, narrow cones relative to the scene
        eye = tuple(rng.uniform([30, 30, 600], [70, 50, 1500]))
    else:             # behind the cloud looking away from most of it (spheres behind the eye), or along the rows
        eye = tuple(rng.uniform([20, 20, -40], [80, 60, 60]))
    w = int(rng.choice([64, 128, 192, 200, 320, 256, 512, 640]))  # (pixel cones from 0.3 to 10 mrad: sharp and coarse lists)
    h = int(rng.choice([48, 96, 200, 384]))
    yaw = float(rng.uniform(-130, -50)) if cam != 4 else float(rng.choice([-90.0, 90.0]) + rng.uniform(-10, 10))
    basis = pt.camera_basis(eye, yaw, float(rng.uniform(-25, 25)), w, h)
    mode, spp, mb = int(seed % 2), int(rng.integers(4, 10)), int(rng.integers(1, 7))
    ref = oracle.render(w, h, spp, spheres=sc, basis=basis, eye=eye, rng_mode=mode, max_bounces=mb, threads=16)
    for v in ((13, 14) if seed % 3 == 0 else (13,)):  # (14: the same kernel with 1024-thread workgroups)
        img, _ = pt.render_frame(w, h, spp, spheres=sc, basis=basis, eye=eye, rng_mode=mode, max_bounces=mb, variant=v)
        neq = int((img.view(np.uint32) != ref.view(np.uint32)).sum())
        floats += img.size
        if neq:
            bad.append({"seed": seed, "variant": v, "floats_different": neq, "n": n, "style": style, "cam": cam, "walls": walls, "size": [w, h], "spp": spp})
    if seed % 25 == 24:
        print(f"seed {seed}: {floats} floats compared, {len(bad)} bad, {time.time() - t0:.0f} s", flush=True)
print(json.dumps({"tool": "primlist_soak", "cases": n_cases, "first_seed": first, "floats_compared": floats, "different": bad[:20], "n_different_cases": len(bad),
                  "fingerprint": pt.build_fingerprint(), "seconds": round(time.time() - t0, 1)}))
