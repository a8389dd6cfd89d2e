#!/usr/bin/env python3
"""Soak of the last-bounce shortcut of the grid kernels (csrc/pt_grid.h PT_V13_LAST_SHORTCUT, EXACTNESS.md A.16): scenes of 100-1500
spheres inside the reference's walls (or five of the six: rays escape through the open side), 0..40 emitting grid spheres (above 32
the rule is off) -- some of them pushed against a wall so that their hit and the wall's are a near tie, some nested in non-emitting
ones, emission from tiny to huge and negative --, sometimes an emitting wall; 2..8 bounces, 4..6 spp.  Variants 13 and 14 against the
CPU oracle, BIT FOR BIT.  Usage: lastbounce_soak.py [n_cases=300] [first_seed=0]"""
import json, os, sys, time
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import __graft_entry__ as ge
pt, oracle = ge.load_package(), ge.load_oracle()
pt.set_device(0)
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad, floats, t0 = [], 0, time.time()
for seed in range(first, first + n_cases):
    rng = np.random.default_rng(9000011 * seed + 3)
    n = int(rng.choice([100, 200, 400, 700, 1000, 1500]))
    sc = pt.scene_random(n, seed=seed, with_walls=True)
    sc["emission"][7:] = 0.0
    n_em = int(rng.choice([0, 1, 3, 8, 16, 31, 32, 33, 40]))
    n_em = min(n_em, n - 7)
    pick = 7 + rng.choice(n - 7, size=n_em, replace=False)
    sc["emission"][pick] = (10.0 ** rng.uniform(-3, 2, size=(n_em, 3))).astype(np.float32) * rng.choice([1.0, 1.0, 1.0, -1.0], size=(n_em, 3)).astype(np.float32)
    for j in pick[: n_em // 3]:  # against a wall: x = 1 (left), x = 99 (right), y = 0 (floor), z = 0 (back) -- tangent within +-1e-3
        r = float(sc["radius"][j])
        wall = int(rng.integers(0, 4))
        p = np.array(sc["pos"][j], dtype=np.float64)
        off = r + rng.uniform(-1e-3, 1e-3)
        if wall == 0: p[0] = 1.0 + off
        elif wall == 1: p[0] = 99.0 - off
        elif wall == 2: p[1] = 0.0 + off
        else: p[2] = 0.0 + off
        sc["pos"][j] = p.astype(np.float32)
    for j in pick[n_em // 3: n_em // 2]:  # nested in a non-emitting sphere of twice the radius at the same centre
        k = 7 + int(rng.integers(0, n - 7))
        if k not in pick:
            sc["pos"][k] = sc["pos"][j]
            sc["radius"][k] = np.float32(2.0 * sc["radius"][j])
    if seed % 4 == 0:
        sc["emission"][int(rng.integers(0, 6))] = rng.uniform(0.0, 1.0, 3).astype(np.float32)  # an emitting wall
    if seed % 5 == 0:
        sc = np.delete(sc, 3)  # no front wall: some paths escape (no colour-variance update for them)
    inside = bool(rng.integers(0, 2))
    eye = tuple(rng.uniform([20, 20, 20], [80, 60, 140])) if inside else tuple(rng.uniform([30, 30, 200], [70, 60, 290]))
    w, h = int(rng.choice([64, 128, 192])), int(rng.choice([48, 64]))
    basis = pt.camera_basis(eye, float(rng.uniform(-120, -60)), float(rng.uniform(-20, 20)), w, h)
    mode, spp, mb = int(seed % 2), int(rng.integers(4, 7)), int(rng.integers(2, 9))
    ref = oracle.render(w, h, spp, spheres=sc, basis=basis, eye=eye, rng_mode=mode, max_bounces=mb, threads=16)
    for v in (13, 14):
        img, _ = pt.render_frame(w, h, spp, spheres=sc, basis=basis, eye=eye, rng_mode=mode, max_bounces=mb, variant=v)
        neq = int((img.view(np.uint32) != ref.view(np.uint32)).sum())
        floats += img.size
        if neq:
            bad.append({"seed": seed, "variant": v, "floats_different": neq, "n": len(sc), "n_em": n_em, "bounces": mb})
    if seed % 25 == 24:
        print(f"seed {seed}: {floats} floats compared, {len(bad)} bad, {time.time() - t0:.0f} s", flush=True)
print(json.dumps({"tool": "lastbounce_soak", "cases": n_cases, "first_seed": first, "floats_compared": floats, "different": bad[:20], "n_different_cases": len(bad),
                  "fingerprint": pt.build_fingerprint(), "seconds": round(time.time() - t0, 1)}))
