#!/usr/bin/env python3
"""Image means of the fast mode vs the exact kernel at 1024 x 1024 x 128 spp for alternative builds (bias hunting: build
variants of pt_fast.hip with tools/build_alt.sh and compare their means; this is how the rounding order of c was found to matter).
Usage: fast_bias.py name...   (cuda-pathtrace_amd/alt/<name>/libptcore.so; "main" = the product library)"""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if sys.argv[1] == "--child":
    import numpy as np
    sys.path.insert(0, root)
    if sys.argv[2] != "main":
        os.environ["PT_LIB_OVERRIDE"] = os.path.join(root, "cuda-pathtrace_amd", "alt", sys.argv[2], "libptcore.so")
    import __graft_entry__ as ge
    pt = ge.load_package(); pt.set_device(0)
    size, spp = 1024, 128
    basis = pt.camera_basis(width=size, height=size)
    f, ms = pt.render_frame(size, size, spp, basis=basis, fast_math=True)
    m = f.reshape(-1, 14).mean(0, dtype=np.float64)
    line = f"{sys.argv[2]:10s} fast {ms:7.3f} ms colour {m[0]:.6f} {m[1]:.6f} {m[2]:.6f} albedo {m[6]:.7f} depth {m[9]:.3f} cvar {m[10]:.6f}"
    if sys.argv[2] == "main":
        e, ms = pt.render_frame(size, size, spp, basis=basis)
        m = e.reshape(-1, 14).mean(0, dtype=np.float64)
        line += f"\n{'exact':10s}      {ms:7.3f} ms colour {m[0]:.6f} {m[1]:.6f} {m[2]:.6f} albedo {m[6]:.7f} depth {m[9]:.3f} cvar {m[10]:.6f}  (se of a colour mean ~ {np.sqrt(m[10] / spp / size / size):.2g})"
    print(line, flush=True)
else:
    for name in sys.argv[1:]:
        subprocess.call([sys.executable, __file__, "--child", name])
