#!/usr/bin/env python3
"""Many-sphere scenes (walls + n random spheres, and the same without walls) at 1024^2 for one or more builds and variants.
Usage: many_ab.py spp name[:variant]...   (name = main or a directory under cuda-pathtrace_amd/alt)"""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if sys.argv[1] == "--child":
    sys.path.insert(0, root)
    name, _, var = sys.argv[3].partition(":")
    var = int(var) if var else None
    if name != "main":
        os.environ["PT_LIB_OVERRIDE"] = os.path.join(root, "cuda-pathtrace_amd", "alt", name, "libptcore.so")
    import __graft_entry__ as ge
    pt = ge.load_package(); pt.set_device(0)
    spp = int(sys.argv[2])
    basis = pt.camera_basis(width=1024, height=1024)
    d_out = pt.DeviceBuffer(1024 * 1024 * 56)
    out = []
    for n in (16, 48, 120, 180, 1000):
        for walls in (True, False):
            scene = pt.scene_random(n, seed=1, with_walls=walls)
            if var is None and n == 1000:
                continue  # the automatic choice there is the grid (tools/cfg4_ab.py)
            r = pt.Renderer(1024, 1024, spp, variant=var)
            d_scene, ns = pt.upload_scene(scene)
            ms = min(r.render(d_out.ptr, d_scene.ptr, ns, basis) for _ in range(3))
            out.append(f"{n}{'c' if walls else 'o'} {ms:7.3f}")
            r.destroy()
    print(f"{sys.argv[3]:10s} spp {spp}: " + " | ".join(out), flush=True)
else:
    for name in sys.argv[2:]:
        subprocess.call([sys.executable, __file__, "--child", sys.argv[1], name])
