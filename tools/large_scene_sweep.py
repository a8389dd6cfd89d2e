import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
pt = ge.load_package(); pt.set_device(0)
basis = pt.camera_basis(width=1024, height=1024)
d_out = pt.DeviceBuffer(1024*1024*14*4)
for n in (600, 800, 850, 860, 900, 1000):
    sc = pt.scene_random(n, seed=1, with_walls=True)
    d_scene, ns = pt.upload_scene(sc)
    row = []
    for v in (6, 8, 10, 13):
        r = pt.Renderer(1024, 1024, 8, variant=v, rng_mode=pt.RNG_PHILOX)
        ms = min(r.render(d_out.ptr, d_scene.ptr, ns, basis) for _ in range(2))
        ki = r.kernel_info(ns)
        row.append((v, round(ms, 2), ki["lds_bytes"], ki["num_vgprs"]))
        r.destroy()
    print(n, row, flush=True)
