#!/usr/bin/env python3
"""Variants 11 (uniform grid, lab), 13 and 14 (pooled grid kernel, 512- and 1024-thread workgroups) against variant 10 (brute force) at full resolution: bit-for-bit comparison of
whole frames on many-sphere scenes with different sphere counts, radii and cameras.  Usage: grid_check.py [spp=8]"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pt = ge.load_lab(); pt.set_device(0)  # grid_header is a lab-library diagnostic
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 8
size = 1024
d_out = pt.DeviceBuffer(size * size * 56)
out = []


def frame(scene, basis, eye, v, mode):
    d_scene, n = pt.upload_scene(scene)
    r = pt.Renderer(size, size, spp, variant=v, rng_mode=mode)
    ms = r.render(d_out.ptr, d_scene.ptr, n, basis, eye)
    img = d_out.download(np.float32, size * size * 14)
    r.destroy()
    return img, ms


rng = np.random.default_rng(2024)
cases = []
for n in (150, 300, 600, 1000, 1500, 2000):
    for walls in (True, False):
        cases.append((f"random{n}_{'walls' if walls else 'open'}", pt.scene_random(n, seed=n, with_walls=walls)))
# radii spread over two decades, clustered positions, touching/overlapping spheres
sc = pt.scene_random(800, seed=5, with_walls=True)
sc["radius"][7:] = np.exp(rng.uniform(np.log(0.05), np.log(6.0), size=len(sc) - 7)).astype(np.float32)
cases.append(("radii_0.05_to_6", sc))
sc = pt.scene_random(800, seed=6, with_walls=True)
sc["pos"][7:] = (np.array([50, 40, 85]) + rng.normal(0, 12, size=(len(sc) - 7, 3))).astype(np.float32)
cases.append(("clustered_gaussian", sc))
sc = pt.scene_random(600, seed=7, with_walls=False)
sc["pos"][:, 1] = 20.0  # coplanar
cases.append(("coplanar_open", sc))
# adversarial geometry: lattice aligned with the cell walls, axis-parallel rays, touching / nested / duplicated spheres
g = np.arange(0, 12)
lat = np.array([(4.0 + 8.0 * x, 4.0 + 8.0 * y, 4.0 + 8.0 * z) for x in g[:10] for y in g[:8] for z in g[:12]], dtype=np.float32)
sc = pt.scene_random(len(lat), seed=8, with_walls=False)
sc["pos"] = lat
sc["radius"] = 4.0  # neighbours touch exactly
cases.append(("lattice_touching", sc))
sc = sc.copy()
sc["radius"] = 2.0
cases.append(("lattice_r2", sc))
sc = pt.scene_random(900, seed=9, with_walls=True)
sc["pos"][300:600] = sc["pos"][7:307]          # nested: same centres, different radii
sc["radius"][300:600] = sc["radius"][7:307] * 0.5
sc[600:893] = sc[7:300]                        # exact duplicates (ties -> first index must win)
cases.append(("nested_and_duplicates", sc))
for scale in (0.01, 100.0):
    sc = pt.scene_random(700, seed=10, with_walls=False)
    sc["pos"] *= scale
    sc["radius"] *= scale
    cases.append((f"scaled_x{scale}", sc))
sc = pt.scene_random(500, seed=11, with_walls=True)
sc["radius"][7:20] = [0.0, -1.0, np.nan, 1e-30, 1e30, np.inf, 0.0, 0.0, 1e-3, 1e-3, 1e-3, 40.0, 60.0]
cases.append(("degenerate_radii", sc))
scaled_cams = {"scaled_x0.01": 0.01, "scaled_x100.0": 100.0}
cams = [((50.0, 52.0, 295.6), -90.0, 0.0), ((50.0, 40.0, 85.0), -60.0, 10.0), ((-150.0, 200.0, 500.0), -55.0, -20.0),
        ((44.0, 36.0, -200.0), 90.0, 0.0)]  # the last one looks straight down a lattice row (axis-parallel centre rays)
for name, scene in cases:
    for ci, (eye, yaw, pitch) in enumerate(cams):
        if name in scaled_cams:
            eye = tuple(c * scaled_cams[name] for c in eye)
        basis = pt.camera_basis(eye, yaw, pitch, size, size)
        mode = ci % 2
        a, ms10 = frame(scene, basis, eye, 10, mode)
        b, ms11 = frame(scene, basis, eye, 11, mode)
        c, ms13 = frame(scene, basis, eye, 13, mode)
        e, ms14 = frame(scene, basis, eye, 14, mode)
        diff = (int((a.view(np.uint32) != b.view(np.uint32)).sum()) + int((a.view(np.uint32) != c.view(np.uint32)).sum()) +
                int((a.view(np.uint32) != e.view(np.uint32)).sum()))
        h = pt.grid_header(scene)
        rec = {"case": name, "camera": ci, "rng": mode, "floats_different": diff, "ms_v10": round(ms10, 2), "ms_v11": round(ms11, 2), "ms_v13": round(ms13, 2), "ms_v14": round(ms14, 2),
               "grid": {"valid": h["valid"], "dims": h["dims"], "n_items": h["n_items"], "n_big": h["n_big"]}}
        out.append(rec)
        print(json.dumps(rec), flush=True)
bad = [r for r in out if r["floats_different"]]
json.dump(out, open(os.path.join("gpurun_out", "grid_check.json"), "w"), indent=1)
print("cases", len(out), "with differences", len(bad))
sys.exit(1 if bad else 0)
