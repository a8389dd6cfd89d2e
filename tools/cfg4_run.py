#!/usr/bin/env python3
"""BASELINE config 4 (1000 random spheres + walls, 1024 x 1024) at a given spp, for profiling:
  cfg4_run.py [spp=16] [open|closed] [reps=2]     (PT_TOOL_VARIANT=<n> in the environment selects a kernel variant)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("PT_LIB_ALT"):
    os.environ["PT_LIB_OVERRIDE"] = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cuda-pathtrace_amd", "alt", os.environ["PT_LIB_ALT"], "libptcore.so")
import __graft_entry__ as ge

pt = ge.load_package()
pt.set_device(0)
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 16
walls = not (len(sys.argv) > 2 and sys.argv[2] == "open")
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
scene = pt.scene_random(1000, seed=1, with_walls=walls)
basis = pt.camera_basis(width=1024, height=1024)
variant = int(os.environ["PT_TOOL_VARIANT"]) if os.environ.get("PT_TOOL_VARIANT") else None
r = pt.Renderer(1024, 1024, spp, variant=variant)
d_scene, n = pt.upload_scene(scene)
d_out = pt.DeviceBuffer(1024 * 1024 * 14 * 4)
ms = [r.render(d_out.ptr, d_scene.ptr, n, basis) for _ in range(reps)]
ki = r.kernel_info(n)
print({"spp": spp, "walls": walls, "variant": ki["variant"], "ms": [round(m, 3) for m in ms],
       "Msamples_per_s": round(1024 * 1024 * spp / min(ms) / 1e3, 1), "fingerprint": pt.build_fingerprint(), "num_vgprs": ki.get("num_vgprs"),
       "samples_per_launch": 1024 * 1024 * spp})
