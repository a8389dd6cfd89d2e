#!/usr/bin/env python3
"""Variant 13 on ROW TILES of the 1000-sphere frame (1024^2 x 256 spp) by number of sample chunks: a tile of few workgroups cannot
use chunk workgroups that only wait for their predecessors.  Usage: cfg4_tile_chunks.py [chunks...]"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import __graft_entry__ as ge
pt = ge.load_package(); pt.set_device(0)
chunks = [int(c) for c in sys.argv[1:]] or [0, 1, 2, 4, 8]
basis = pt.camera_basis(width=1024, height=1024)
for walls in (True, False):
    scene = pt.scene_random(1000, seed=1, with_walls=walls)
    d_scene, n = pt.upload_scene(scene)
    for rows in ((0, 1024), (512, 1024), (512, 768), (512, 640), (512, 576)):
        d_out = pt.DeviceBuffer((rows[1] - rows[0]) * 1024 * 56)
        res = []
        for c in chunks:
            r = pt.Renderer(1024, 1024, 256, chunks=c, row_begin=rows[0], row_end=rows[1])
            ms = min(r.render(d_out.ptr, d_scene.ptr, n, basis) for _ in range(2))
            res.append(f"{c}: {ms:7.2f}")
            r.destroy()
        print(f"{'closed' if walls else 'open'} rows {rows[0]}..{rows[1]}  ms by chunks  " + "  ".join(res), flush=True)
        d_out.free()
