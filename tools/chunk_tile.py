#!/usr/bin/env python3
"""Sample chunking of variant 6 on row tiles of the headline frame: kernel ms by chunk count.  Usage: chunk_tile.py rows..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pt = ge.load_package(); pt.set_device(0)
basis = pt.camera_basis(width=1024, height=1024)
d_scene, n = pt.upload_scene(pt.scene_cornell())
d_out = pt.DeviceBuffer(1024 * 1024 * 56)
for rows in [int(x) for x in sys.argv[1:]] or [128, 192, 256]:
    for v in [int(x) for x in os.environ.get("PT_TOOL_VARIANTS", "6").split(",")]:
        out = []
        for chunks in (1, 2, 3, 4, 5, 6, 8, 12, 16):
            r = pt.Renderer(1024, 1024, 1024, variant=v, chunks=chunks, row_end=rows)
            ms = sorted(r.render(d_out.ptr, d_scene.ptr, n, basis) for _ in range(4))
            out.append(f"{chunks}: {ms[0]:.3f}")
            r.destroy()
        print(f"rows {rows} ({rows / 64:.1f} one-lane waves per SIMD) variant {v}, ms by chunks: " + " | ".join(out), flush=True)
