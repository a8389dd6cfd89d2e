#!/usr/bin/env python3
"""Fast-mode kernel time at 1024^2 x 256 spp for alternative builds.  Usage: fast_ab.py name..."""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if sys.argv[1] == "--child":
    sys.path.insert(0, root)
    if sys.argv[2] != "main":
        os.environ["PT_LIB_OVERRIDE"] = os.path.join(root, "cuda-pathtrace_amd", "alt", sys.argv[2], "libptcore.so")
    import __graft_entry__ as ge
    pt = ge.load_package(); pt.set_device(0)
    basis = pt.camera_basis(width=1024, height=1024)
    r = pt.Renderer(1024, 1024, 256, fast_math=True, persist_rng=False)
    d_scene, n = pt.upload_scene(pt.scene_cornell())
    d_out = pt.DeviceBuffer(1024 * 1024 * 56)
    ms = min(r.render(d_out.ptr, d_scene.ptr, n, basis) for _ in range(4))
    print(f"{sys.argv[2]:8s} fast 1024^2x256: {ms:7.3f} ms  vgpr {r.kernel_info(n)['num_vgprs']} scratch {r.kernel_info(n)['scratch_bytes']}", flush=True)
else:
    for name in sys.argv[1:]:
        subprocess.call([sys.executable, __file__, "--child", name])
