#!/usr/bin/env python3
"""A/B alternative builds (cuda-pathtrace_amd/alt/<name>/libptcore.so) on the full frame and on
the 1/8 row tile (low occupancy).  Usage: ab_tiles.py variant name1 name2 ...   (runs each in a subprocess)"""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 2 and sys.argv[1] == "--child":
    import ctypes, numpy as np
    sys.path.insert(0, root)
    os.environ["PT_LIB_OVERRIDE"] = sys.argv[2]
    import __graft_entry__ as ge
    pt = ge.load_package()
    pt.set_device(0)
    v = int(sys.argv[3])
    basis = pt.camera_basis(width=1024, height=1024)
    d_scene, n = pt.upload_scene(pt.scene_cornell())
    res = []
    for rows, spp in (((0, 1024), 256), ((0, 128), 1024), ((0, 256), 1024)):
        r = pt.Renderer(1024, 1024, spp, variant=v, row_begin=rows[0], row_end=rows[1], persist_rng=False)
        d_out = pt.DeviceBuffer((rows[1] - rows[0]) * 1024 * 56)
        ms = min(r.render(d_out.ptr, d_scene.ptr, n, basis) for _ in range(3))
        ki = r.kernel_info(n)
        res.append(f"rows {rows[1]-rows[0]:4d} x {spp:4d}spp: {ms:8.3f} ms {(rows[1]-rows[0])*1024*spp/ms/1e3:8.0f} Ms/s")
        r.destroy(); d_out.free()
    print(f"{os.path.basename(os.path.dirname(sys.argv[2])):10s} vgpr {ki['num_vgprs']:3d} scratch {ki['scratch_bytes']:3d} | " + " | ".join(res), flush=True)
else:
    v = sys.argv[1]
    for name in sys.argv[2:]:
        lib = os.path.join(root, "cuda-pathtrace_amd", "alt", name, "libptcore.so") if name != "main" else os.path.join(root, "cuda-pathtrace_amd", "libptcore.so")
        subprocess.call([sys.executable, __file__, "--child", lib, v])
