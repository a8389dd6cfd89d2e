#!/usr/bin/env python3
"""Variant 6 (one lane per pixel) vs 8 (four lanes per pixel) per tile size and generator: is the automatic policy right?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pt = ge.load_package(); pt.set_device(0)
basis = pt.camera_basis(width=1024, height=1024)
d_scene, n = pt.upload_scene(pt.scene_cornell())
for rng in (0, 1):
    for rows in (1024, 512, 384, 256, 128):
        out = []
        for v in (6, 8, None):
            r = pt.Renderer(1024, 1024, 1024, variant=v, rng_mode=rng, row_begin=0, row_end=rows, persist_rng=True)
            d_out = pt.DeviceBuffer(rows * 1024 * 56)
            ms = min(r.render(d_out.ptr, d_scene.ptr, n, basis) for _ in range(3))
            out.append(f"v{v if v is not None else 'auto(' + str(r.kernel_info(n)['variant']) + ')'} {ms:7.3f}")
            r.destroy(); d_out.free()
        print(("xorwow" if rng == 0 else "philox"), f"rows {rows:4d}:", " | ".join(out), flush=True)
