#!/usr/bin/env python3
"""Headline frame (1024^2 x 1024 spp Cornell box) kernel ms for alternative builds.  Usage: headline_ab.py name[:variant]..."""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if sys.argv[1] == "--child":
    sys.path.insert(0, root)
    name, _, var = sys.argv[2].partition(":")
    var = int(var) if var else None
    if name != "main":
        os.environ["PT_LIB_OVERRIDE"] = os.path.join(root, "cuda-pathtrace_amd", "alt", name, "libptcore.so")
    import __graft_entry__ as ge
    pt = ge.load_package(); pt.set_device(0)
    basis = pt.camera_basis(width=1024, height=1024)
    d_scene, n = pt.upload_scene(pt.scene_cornell())
    d_out = pt.DeviceBuffer(1024 * 1024 * 56)
    out = []
    for rows in (1024, 256, 128):
        r = pt.Renderer(1024, 1024, 1024, variant=var, row_end=rows)
        ms = sorted(r.render(d_out.ptr, d_scene.ptr, n, basis) for _ in range(4))
        ki = r.kernel_info(n)
        out.append(f"{rows} rows: {ms[0]:.3f} ms (v{ki['variant']}, {ki['num_vgprs']} vgprs)")
        r.destroy()
    print(f"{sys.argv[2]:10s} " + " | ".join(out), flush=True)
else:
    for name in sys.argv[1:]:
        subprocess.call([sys.executable, __file__, "--child", name])
