import os, sys, ctypes
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
os.environ["PT_LIB_OVERRIDE"] = os.path.join(root, "cuda-pathtrace_amd", "alt", "griddbg", "libptcore.so")
import numpy as np
import __graft_entry__ as ge
pt = ge.load_package(); pt.set_device(0)
basis = pt.camera_basis(width=256, height=256)
d_out = pt.DeviceBuffer(256*256*56)
cnt = (ctypes.c_ulonglong * 8)()
for n, walls in ((1000, True), (1000, False)):
    sc = pt.scene_random(n, seed=1, with_walls=walls)
    d_scene, ns = pt.upload_scene(sc)
    r = pt.Renderer(256, 256, 4, variant=11)
    pt.lib.pt_debug_grid_counters(cnt, 1)
    ms = r.render(d_out.ptr, d_scene.ptr, ns, basis)
    pt.lib.pt_debug_grid_counters(cnt, 1)
    c = list(cnt)
    print(n, walls, "ms", round(ms, 3), dict(rays=c[0], iters=c[1], tests=c[2], ambiguous=c[4], unsure=c[5], wave_calls=c[3], wave_literal=c[6], wave_iters=c[7]),
          "iters/ray", round(c[1]/max(c[0],1), 2), "wave iters per wave call", round(c[7]/max(c[3],1), 2), "literal wave frac", round(c[6]/max(c[3],1), 4))
    r.destroy()
    r = pt.Renderer(256, 256, 4, variant=10)
    print("   v10 ms", round(r.render(d_out.ptr, d_scene.ptr, ns, basis), 3))
    r.destroy()
