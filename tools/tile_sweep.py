import sys, os
sys.path.insert(0,'.')
import __graft_entry__ as ge
pt=ge.load_package(); pt.set_device(0)
basis=pt.camera_basis(width=1024,height=1024); d_scene,n=pt.upload_scene(pt.scene_cornell())
for rows in (1024,512,384,256,192,128,64):
    out=[]
    for v in (6,9,8):
        r=pt.Renderer(1024,1024,1024,variant=v,row_begin=0,row_end=rows); d=pt.DeviceBuffer(rows*1024*56)
        ms=min(r.render(d.ptr,d_scene.ptr,n,basis) for _ in range(3)); r.destroy(); d.free(); out.append(ms)
    print(f"rows {rows:4d} ({rows*1024/64/1024:.1f} waves/SIMD): v6 {out[0]:7.3f} ms  v9(S=2) {out[1]:7.3f} ms  v8(S=4) {out[2]:7.3f} ms")
