#!/usr/bin/env python3
"""Sample chunking at the headline configuration: kernel ms for 1 (off), 2, 4, 8, 16 chunks, both generators.
Usage: chunk_sweep.py [reps=4]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pt = ge.load_package(); pt.set_device(0)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
basis = pt.camera_basis(width=1024, height=1024)
d_scene, n = pt.upload_scene(pt.scene_cornell())
d_out = pt.DeviceBuffer(1024 * 1024 * 56)
out = {"fingerprint": pt.build_fingerprint()}
for rows, name in ((1024, "full"), (512, "half")):
    for rng in (0, 1):
        for chunks in (1, 2, 4, 8, 16):
            r = pt.Renderer(1024, 1024, 1024, rng_mode=rng, variant=6, chunks=chunks, row_end=rows)
            ms = sorted(r.render(d_out.ptr, d_scene.ptr, n, basis) for _ in range(reps))
            ki = r.kernel_info(n)
            r.destroy()
            out[f"{name}_rng{rng}_chunks{chunks}"] = {"ms_min": round(ms[0], 3), "ms_median": round(ms[len(ms) // 2], 3), "grid_blocks": ki["grid_blocks"]}
            print(name, "rng", rng, "chunks", chunks, out[f"{name}_rng{rng}_chunks{chunks}"], flush=True)
json.dump(out, open(os.path.join("gpurun_out", "chunk_sweep.json"), "w"), indent=1)
