#!/usr/bin/env python3
"""Philox: variant 6 against variant 8 by tile size (the automatic choice is 8 from 4 spp up; at the full frame 6 is 1.6 % ahead,
from half a frame down 8 is 2-15 % ahead: policy left as it is)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pt=ge.load_package(); pt.set_device(0)
basis=pt.camera_basis(width=1024,height=1024)
d_scene,n=pt.upload_scene(pt.scene_cornell())
d_out=pt.DeviceBuffer(1024*1024*56)
for spp in (64,1024):
  for rows in (1024,512,256,128):
    res=[]
    for v in (6,8):
        r=pt.Renderer(1024,1024,spp,rng_mode=1,variant=v,row_begin=0,row_end=rows)
        ms=min(r.render(d_out.ptr,d_scene.ptr,n,basis) for _ in range(3))
        res.append(ms); r.destroy()
    print(f"philox spp {spp} rows {rows}: v6 {res[0]:.3f} ms, v8 {res[1]:.3f} ms",flush=True)
