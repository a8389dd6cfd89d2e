#!/usr/bin/env python3
"""Statistics of the toleranced fast mode against the exact kernel at equal seeds (sets the tolerances of
tests/test_fast_mode_gpu.py).  Usage: fast_stats.py"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pt = ge.load_package(); pt.set_device(0)
out = {}
for size, spp in ((256, 1), (256, 64), (256, 256), (512, 1024)):
    basis = pt.camera_basis(width=size, height=size)
    e, ms_e = pt.render_frame(size, size, spp, basis=basis)
    f, ms_f = pt.render_frame(size, size, spp, basis=basis, fast_math=True)
    same_alb = np.all(e[..., 6:9] == f[..., 6:9], axis=-1)
    nd = np.abs(e[..., 3:6] - f[..., 3:6]).max(-1)
    cd = np.abs(e[..., :3] - f[..., :3]).max(-1)
    rd = np.abs(e[..., 9] - f[..., 9]) / e[..., 9]
    q = lambda a, p: float(np.quantile(a, p))
    rec = {"ms_exact": round(ms_e, 3), "ms_fast": round(ms_f, 3), "albedo_differs_share": float((~same_alb).mean()),
           "normal_absdiff": {"median": q(nd, .5), "p99": q(nd, .99), "p999": q(nd, .999), "max": float(nd.max())},
           "colour_absdiff": {"median": q(cd, .5), "p90": q(cd, .9), "p99": q(cd, .99), "max": float(cd.max()), "share_gt_1e-4": float((cd > 1e-4).mean())},
           "depth_reldiff": {"median": q(rd, .5), "p99": q(rd, .99), "p999": q(rd, .999), "max": float(rd.max())},
           "means_exact": [float(x) for x in e.reshape(-1, 14).mean(0, dtype=np.float64)],
           "means_fast": [float(x) for x in f.reshape(-1, 14).mean(0, dtype=np.float64)]}
    out[f"{size}x{size}x{spp}"] = rec
    print(size, spp, json.dumps(rec), flush=True)
json.dump(out, open(os.path.join("gpurun_out", "fast_stats.json"), "w"), indent=1)
