#!/usr/bin/env python3
"""Config 4 (1000 spheres, 1024^2 x 256 spp, closed and open) by number of sample chunks (variant 13).  With path regeneration a lane
works through its own samples at its own pace; a chunk ends when its SLOWEST lane is done, so short chunks thin the waves out --
most where path lengths vary most (open scenes).  Usage: cfg4_chunks.py [spp=256] [chunks...]"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import __graft_entry__ as ge
pt = ge.load_package(); pt.set_device(0)
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 256
chunks = [int(c) for c in sys.argv[2:]] or [0, 1, 2, 4, 8, 16]
basis = pt.camera_basis(width=1024, height=1024)
for walls in (True, False):
    scene = pt.scene_random(1000, seed=1, with_walls=walls)
    d_scene, n = pt.upload_scene(scene)
    d_out = pt.DeviceBuffer(1024 * 1024 * 56)
    for c in chunks:
        r = pt.Renderer(1024, 1024, spp, variant=13, chunks=c)
        ms = min(r.render(d_out.ptr, d_scene.ptr, n, basis) for _ in range(3))
        info = r.kernel_info(n)
        print(f"{'closed' if walls else 'open'} spp {spp} chunks requested {c} (used {info.get('chunks')}): {ms:8.3f} ms", flush=True)
        r.destroy()
