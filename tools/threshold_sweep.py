#!/usr/bin/env python3
"""Where should scenes switch from the key-based screen (LDS image) to the many-sphere loop (scalar loads)?
Compares the main build with an alternative build that has a low PT_SCREEN_MAX_SPHERES
(tools/build_alt.sh t9 -DPT_SCREEN_MAX_SPHERES=9) on random scenes of 10..64 spheres.
Usage: threshold_sweep.py [alt_name=t9]"""
import os, subprocess, sys, json
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 2 and sys.argv[1] == "--child":
    sys.path.insert(0, root)
    os.environ["PT_LIB_OVERRIDE"] = sys.argv[2]
    import numpy as np
    import __graft_entry__ as ge
    pt = ge.load_package()
    pt.set_device(0)
    basis = pt.camera_basis(width=1024, height=1024)
    d_out = pt.DeviceBuffer(1024 * 1024 * 56)
    out = {}
    for n in (10, 12, 16, 24, 32, 48, 64):
        for walls in (True, False):
            sc = pt.scene_random(n, seed=7, with_walls=walls)
            d_scene, ns = pt.upload_scene(sc)
            for v in (6, None):
                r = pt.Renderer(1024, 1024, 16, variant=v)
                ms = min(r.render(d_out.ptr, d_scene.ptr, ns, basis) for _ in range(3))
                used = r.kernel_info(ns)["variant"]
                r.destroy()
                out[f"{n}_{'walls' if walls else 'open'}_{'auto' if v is None else v}"] = (round(ms, 3), used)
    print(json.dumps(out))
else:
    alt = sys.argv[1] if len(sys.argv) > 1 else "t9"
    libs = {"main": os.path.join(root, "cuda-pathtrace_amd", "libptcore.so"),
            alt: os.path.join(root, "cuda-pathtrace_amd", "alt", alt, "libptcore.so")}
    res = {k: json.loads(subprocess.check_output([sys.executable, __file__, "--child", p]).decode().strip().split("\n")[-1]) for k, p in libs.items()}
    for key in res["main"]:
        print(f"{key:22s} main {res['main'][key]}  {alt} {res[alt][key]}", flush=True)
