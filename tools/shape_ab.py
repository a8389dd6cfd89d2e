#!/usr/bin/env python3
"""tools/shape_sweep.py's frames for alternative builds.  Usage: shape_ab.py name..."""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if sys.argv[1] == "--child":
    sys.path.insert(0, root)
    if sys.argv[2] != "main":
        os.environ["PT_LIB_OVERRIDE"] = os.path.join(root, "cuda-pathtrace_amd", "alt", sys.argv[2], "libptcore.so")
    import __graft_entry__ as ge
    pt = ge.load_package(); pt.set_device(0)
    d_scene, n = pt.upload_scene(pt.scene_cornell())
    out = []
    for size, spp in ((512, 4096), (1024, 1024), (2048, 256), (4096, 64)):
        basis = pt.camera_basis(width=size, height=size)
        r = pt.Renderer(size, size, spp, variant=6)
        d_out = pt.DeviceBuffer(size * size * 56)
        ms = min(r.render(d_out.ptr, d_scene.ptr, n, basis) for _ in range(3))
        out.append(f"{size}^2x{spp} {ms:7.3f}")
        r.destroy(); d_out.free()
    print(f"{sys.argv[2]:8s} " + " | ".join(out), flush=True)
else:
    for name in sys.argv[1:]:
        subprocess.call([sys.executable, __file__, "--child", name])
