#!/usr/bin/env python3
"""The headline kernel on frames of equal sample count (2^30) but different shape: is the time per sample a property of
the kernel or of how many workgroups the frame makes?  Usage: shape_sweep.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pt = ge.load_package(); pt.set_device(0)
d_scene, n = pt.upload_scene(pt.scene_cornell())
for size, spp in ((512, 4096), (1024, 1024), (2048, 256), (4096, 64), (8192, 16)):
    basis = pt.camera_basis(width=size, height=size)
    r = pt.Renderer(size, size, spp, variant=6)
    d_out = pt.DeviceBuffer(size * size * 56)
    ms = min(r.render(d_out.ptr, d_scene.ptr, n, basis) for _ in range(3))
    ki = r.kernel_info(n)
    print(f"{size}x{size}x{spp}: {ms:.3f} ms, {size * size * spp / ms / 1e6:.2f} Gsamples/s, {size * size // 256} workgroups, vgpr {ki['num_vgprs']}", flush=True)
    r.destroy(); d_out.free()
