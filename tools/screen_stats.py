#!/usr/bin/env python3
"""How often, and why, lanes of the headline kernel leave the screened nearest-hit search for the literal loop
(instrumentation build: tools/build_alt.sh sstats -DPT_SCREEN_STATS).  Usage: screen_stats.py [spp=16]"""
import ctypes, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["PT_LIB_OVERRIDE"] = os.path.join(root, "cuda-pathtrace_amd", "alt", "sstats", "libptcore.so")
sys.path.insert(0, root)
import __graft_entry__ as ge
pt = ge.load_package(); pt.set_device(0)
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 16
basis = pt.camera_basis(width=1024, height=1024)
scene = pt.scene_cornell()
r = pt.Renderer(1024, 1024, spp)
d_scene, n = pt.upload_scene(scene)
d_out = pt.DeviceBuffer(1024 * 1024 * 56)
st = (ctypes.c_ulonglong * 8)()
pt.lib.pt_debug_screen_stats(st, 1)
ms = r.render(d_out.ptr, d_scene.ptr, n, basis)
pt.lib.pt_debug_screen_stats(st, 1)
wb, wany, amb, unsure, tie, lim, bad, notgood = [st[i] for i in range(8)]
print(f"variant {r.kernel_info(n)['variant']}, {ms:.1f} ms instrumented; wave-bounces {wb}; with a lane on the literal loop {wany} = {100 * wany / wb:.2f} %; "
      f"lanes: {amb} = {1e6 * amb / (wb * 64):.0f} per million (unsure {unsure}, near tie {tie}, at the limit {lim}, exact step doubted {bad}, rejected {notgood})")
