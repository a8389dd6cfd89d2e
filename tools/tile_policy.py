#!/usr/bin/env python3
"""Where does the four-lane kernel (variant 8) beat the one-lane kernel (variant 6) on row tiles of the headline frame?
Kernel ms per tile height, both generators.  Usage: tile_policy.py [rows...]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pt = ge.load_package(); pt.set_device(0)
rows_list = [int(x) for x in sys.argv[1:]] or [64, 96, 128, 160, 192, 224, 256, 320, 384, 512, 768, 1024]
basis = pt.camera_basis(width=1024, height=1024)
d_scene, n = pt.upload_scene(pt.scene_cornell())
d_out = pt.DeviceBuffer(1024 * 1024 * 56)
out = {"fingerprint": pt.build_fingerprint(), "device": pt.device_info()}
for rng in (0, 1):
    for rows in rows_list:
        rec = {}
        for v in (6, 8, 9, None):
            r = pt.Renderer(1024, 1024, 1024, variant=v, row_end=rows, rng_mode=rng)
            ms = sorted(r.render(d_out.ptr, d_scene.ptr, n, basis) for _ in range(3))
            rec["auto=%d" % r.kernel_info(n)["variant"] if v is None else "v%d" % v] = round(ms[0], 3)
            r.destroy()
        waves_per_simd = rows * 1024 / 64 / (out["device"]["compute_units"] * 4)
        out[f"rng{rng}_rows{rows}"] = rec
        print(f"rng {rng} rows {rows:5d} ({waves_per_simd:5.2f} one-lane waves per SIMD): {rec}", flush=True)
json.dump(out, open(os.path.join("gpurun_out", "tile_policy.json"), "w"), indent=1)
