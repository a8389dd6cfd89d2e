#!/usr/bin/env python3
"""A/B two builds of libptcore.so on the headline frame with raw ctypes (works across ABI additions).
Usage: ab_raw.py lib_a.so lib_b.so [reps]   (each library is timed in its own subprocess, alternating)"""
import ctypes, os, subprocess, sys
if sys.argv[1] == "--child":
    L = ctypes.CDLL(sys.argv[2])
    vp = ctypes.c_void_p
    L.pt_malloc.argtypes = [ctypes.POINTER(vp), ctypes.c_size_t]
    L.pt_memcpy_h2d.argtypes = [vp, vp, ctypes.c_size_t]
    L.pt_renderer_create.argtypes = [ctypes.c_int] * 4 + [vp, ctypes.POINTER(vp)]
    L.pt_renderer_render.argtypes = [vp, vp, vp, ctypes.c_int, vp, vp, ctypes.POINTER(ctypes.c_float)]
    L.pt_camera_basis.argtypes = [vp, ctypes.c_float, ctypes.c_float, ctypes.c_int, ctypes.c_int, vp]
    assert L.pt_set_device(0) == 0
    host = (ctypes.c_float * 90)()
    L.pt_scene_cornell(host)
    d_s, d_o, r = vp(), vp(), vp()
    L.pt_malloc(ctypes.byref(d_s), 360); L.pt_memcpy_h2d(d_s, host, 360)
    L.pt_malloc(ctypes.byref(d_o), 1024 * 1024 * 56)
    eye = (ctypes.c_float * 3)(50.0, 52.0, 295.6)
    basis = (ctypes.c_float * 12)()
    L.pt_camera_basis(eye, -90.0, 0.0, 1024, 1024, basis)
    assert L.pt_renderer_create(1024, 1024, 1024, 8, None, ctypes.byref(r)) == 0
    ms = ctypes.c_float()
    out = []
    for _ in range(int(sys.argv[3])):
        assert L.pt_renderer_render(r, d_o, d_s, 9, basis, eye, ctypes.byref(ms)) == 0
        out.append(ms.value)
    print(os.path.basename(os.path.dirname(sys.argv[2])), "min %.3f  all %s" % (min(out), " ".join("%.2f" % m for m in out)), flush=True)
else:
    reps = sys.argv[3] if len(sys.argv) > 3 else "6"
    for _ in range(2):
        for lib in sys.argv[1:3]:
            subprocess.call([sys.executable, __file__, "--child", os.path.abspath(lib), reps])
