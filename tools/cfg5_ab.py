#!/usr/bin/env python3
"""BASELINE config 5 (512^2 x 4 spp x 8 bounces per frame): kernel ms and stream ms per frame for variants / generators / builds.
Usage: cfg5_ab.py name[:variant]...   (name = main or a directory under cuda-pathtrace_amd/alt)"""
import os, subprocess, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if sys.argv[1] == "--child":
    sys.path.insert(0, root)
    name, _, var = sys.argv[2].partition(":")
    var = int(var) if var else None
    if name != "main":
        os.environ["PT_LIB_OVERRIDE"] = os.path.join(root, "cuda-pathtrace_amd", "alt", name, "libptcore.so")
    import __graft_entry__ as ge
    pt = ge.load_package(); pt.set_device(0)
    basis = pt.camera_basis(width=512, height=512)
    d_scene, n = pt.upload_scene(pt.scene_cornell())
    d_out = pt.DeviceBuffer(512 * 512 * 56)
    out = []
    for rng in (0, 1):
        r = pt.Renderer(512, 512, 4, variant=var, max_bounces=8, rng_mode=rng)
        ms = sorted(r.render(d_out.ptr, d_scene.ptr, n, basis) for _ in range(30))
        pt.check(pt.lib.pt_device_synchronize())
        t = time.perf_counter()
        for _ in range(300):
            r.enqueue(d_out.ptr, d_scene.ptr, n, basis)
        pt.check(pt.lib.pt_device_synchronize())
        wall = (time.perf_counter() - t) / 300 * 1e3
        ki = r.kernel_info(n)
        out.append(f"rng {rng}: kernel min {ms[0]:.4f} med {ms[15]:.4f} ms, stream {wall:.4f} ms/frame (variant {ki['variant']}, {ki['num_vgprs']} vgprs)")
        r.destroy()
    print(f"{sys.argv[2]:10s} " + " | ".join(out), flush=True)
else:
    for name in sys.argv[1:]:
        subprocess.call([sys.executable, __file__, "--child", name])
