#!/usr/bin/env python3
"""When do the headline kernel's waves start and finish (instrumentation build: tools/build_alt.sh wtime -DPT_WAVE_TIMING)?
Usage: wave_timing.py"""
import ctypes, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["PT_LIB_OVERRIDE"] = os.path.join(root, "cuda-pathtrace_amd", "alt", "wtime", "libptcore.so")
sys.path.insert(0, root)
import __graft_entry__ as ge
pt = ge.load_package(); pt.set_device(0)
d_scene, n = pt.upload_scene(pt.scene_cornell())
st = (ctypes.c_ulonglong * 8)()
for size, spp in ((512, 4096), (1024, 1024), (2048, 256)):
    basis = pt.camera_basis(width=size, height=size)
    r = pt.Renderer(size, size, spp, variant=6)
    d_out = pt.DeviceBuffer(size * size * 56)
    r.render(d_out.ptr, d_scene.ptr, n, basis)
    pt.lib.pt_debug_wave_timing(st, 1)
    ms = r.render(d_out.ptr, d_scene.ptr, n, basis)
    pt.lib.pt_debug_wave_timing(st, 1)
    s0, s1, e0, e1, life, cnt, esum = [st[i] for i in range(7)]
    tick = 1e-5  # ms per tick of the 100 MHz counter
    span = (e1 - s0) * tick
    print(f"{size}^2x{spp}: kernel {ms:.2f} ms; waves {cnt}; first start..last start {(s1 - s0) * tick:.2f} ms; first end {(e0 - s0) * tick:.2f} ms, last end {span:.2f} ms; "
          f"mean wave lifetime {life / cnt * tick:.2f} ms", flush=True)
    r.destroy(); d_out.free()
