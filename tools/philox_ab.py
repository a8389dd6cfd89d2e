import os, subprocess, sys
root='/root/repo'
if sys.argv[1]=="--child":
    sys.path.insert(0, root)
    name=sys.argv[2]
    if name!="main": os.environ["PT_LIB_OVERRIDE"]=os.path.join(root,"cuda-pathtrace_amd","alt",name,"libptcore.so")
    import __graft_entry__ as ge
    pt=ge.load_package(); pt.set_device(0)
    basis=pt.camera_basis(width=1024,height=1024)
    d_scene,n=pt.upload_scene(pt.scene_cornell()); d_out=pt.DeviceBuffer(1024*1024*56)
    out=[]
    for rows in (1024,512,256):
        r=pt.Renderer(1024,1024,1024,rng_mode=1,row_end=rows)
        ms=sorted(r.render(d_out.ptr,d_scene.ptr,n,basis) for _ in range(4)); ki=r.kernel_info(n)
        out.append(f"{rows}: {ms[0]:.3f} (v{ki['variant']}, {ki['num_vgprs']} vgprs, scratch {ki['scratch_bytes']})"); r.destroy()
    r=pt.Renderer(512,512,4,rng_mode=1,max_bounces=8,variant=6); d2=pt.DeviceBuffer(512*512*56); b2=pt.camera_basis(width=512,height=512)
    ms=sorted(r.render(d2.ptr,d_scene.ptr,n,b2) for _ in range(30)); out.append(f"cfg5 v6 philox {ms[0]:.4f}")
    print(f"{name:6s} "+" | ".join(out),flush=True)
else:
    for name in sys.argv[1:]: subprocess.call([sys.executable,__file__,"--child",name])
