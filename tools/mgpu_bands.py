#!/usr/bin/env python3
"""Pipelined exchange inside a frame (pt_mgpu_opts.bands): what banding costs the render and what it hides of the exchange, on
whatever devices the box has.  One GPU: ranks share device 0 and every tile goes through the exchange (force_exchange; peer
copies, or RCCL self send/recv with one rank).  Usage: mgpu_bands.py [width=4096] [rows per rank=512] [spp=64] [ranks=1]"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pt = ge.load_package(); pt.set_device(0)
width = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 512
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 64
ranks = int(sys.argv[4]) if len(sys.argv) > 4 else 1
ndev = pt.device_count()
devices = list(range(ranks)) if ndev >= ranks and ranks > 1 else [0] * ranks
height = rows * ranks
basis = pt.camera_basis(width=width, height=height)
d_scene, n = pt.upload_scene(pt.scene_cornell())
d_out = pt.DeviceBuffer(width * height * 56)
ref = None
for gather, gname in ((pt.GATHER_PEER_COPY, "copy"), (pt.GATHER_RCCL, "rccl")):
    if gather == pt.GATHER_RCCL and len(set(devices)) != len(devices):
        continue
    for bands in (1, 2, 4, 8):
        m = pt.MultiRenderer(devices, width, height, spp, force_exchange=True, gather=gather, bands=bands, persist_rng=False)
        runs = []
        for _ in range(5):
            wall = m.render(d_out.ptr, d_scene.ptr, n, basis)
            fs = m.frame_stats()
            runs.append((wall, fs["render_ms"], fs["exposed_ms"]))
        img = d_out.download(np.float32, (height, width, 14))
        if ref is None:
            ref = img
        same = bool(np.array_equal(img.view(np.uint32), ref.view(np.uint32)))
        best = min(runs[1:])
        print(json.dumps({"devices": devices, "exchange": m.backend(), "bands": bands, "tile_MB": round(rows * width * 56 / 1e6, 1),
                          "wall_ms": round(best[0], 3), "render_ms": round(best[1], 3), "exposed_ms": round(best[2], 3),
                          "frame_equals_bands1": same}), flush=True)
        m.destroy()
