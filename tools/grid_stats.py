#!/usr/bin/env python3
"""What the grid walk does per wave (instrumentation build: tools/build_alt.sh stats -DPT_GRID_STATS).
Usage: grid_stats.py [spp=8] [variant=11] [alt build name=stats]"""
import ctypes, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["PT_LIB_OVERRIDE"] = os.path.join(root, "cuda-pathtrace_amd", "alt", sys.argv[3] if len(sys.argv) > 3 else "stats", "libptcore.so")
sys.path.insert(0, root)
import __graft_entry__ as ge
pt = ge.load_package(); pt.set_device(0)
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 8
variant = int(sys.argv[2]) if len(sys.argv) > 2 else 11
basis = pt.camera_basis(width=1024, height=1024)
for walls in (True, False):
    scene = pt.scene_random(1000, seed=1, with_walls=walls)
    r = pt.Renderer(1024, 1024, spp, variant=variant)
    d_scene, n = pt.upload_scene(scene)
    d_out = pt.DeviceBuffer(1024 * 1024 * 56)
    st = (ctypes.c_ulonglong * 8)()
    pt.lib.pt_debug_grid_stats(st, 1)
    if hasattr(pt.lib, "pt_debug_grid_hist"):
        pt.lib.pt_debug_grid_hist((ctypes.c_ulonglong * 256)(), 1)
    ms = r.render(d_out.ptr, d_scene.ptr, n, basis)
    pt.lib.pt_debug_grid_stats(st, 1)
    walks, trips, lanes_t, rounds, lanes_s, entered, amb = [st[i] for i in range(7)]
    print(f"{'closed' if walls else 'open'}: {ms:.2f} ms (instrumented); per wave walk: {trips / walks:.1f} test trips with {lanes_t / max(trips, 1):.1f} lanes testing, "
          f"{rounds / walks:.1f} step rounds with {lanes_s / max(rounds, 1):.1f} lanes stepping; lanes entering {entered / walks:.1f}; per entering lane: "
          f"{lanes_t / max(entered, 1):.1f} tests, {lanes_s / max(entered, 1):.1f} steps; ambiguous lanes {amb / max(entered, 1) * 100:.3f} %; waves with a lane on the brute-force path {st[7] / walks * 100:.2f} %")
    if len(sys.argv) > 3 and "amb" in sys.argv[3]:
        print(f"   ambiguous lanes {st[6]} of {entered}: unsure {st[1]}, near tie {st[2]}, at the limit {st[3]}, exact step doubted {st[4]}, exact step rejects the estimate's hit {st[7]}")
    print("   raw", [st[i] for i in range(8)], "wave-bounces (closed) =", 1024 * 1024 * spp * 5 // 64)
    if hasattr(pt.lib, "pt_debug_grid_hist"):
        hh = (ctypes.c_ulonglong * 256)()
        pt.lib.pt_debug_grid_hist(hh, 1)
        for k, name in enumerate(("tests per ray", "steps per ray", "test trips per wave walk", "step rounds per wave walk")):
            v = [hh[64 * k + i] for i in range(64)]
            tot = max(sum(v), 1)
            mean = sum(i * x for i, x in enumerate(v)) / tot
            cum, q = 0, {}
            for i, x in enumerate(v):
                cum += x
                for f in (0.5, 0.9, 0.99):
                    if f not in q and cum >= f * tot:
                        q[f] = i
            print(f"   hist {name}: mean {mean:.2f} p50 {q.get(0.5)} p90 {q.get(0.9)} p99 {q.get(0.99)} | " + " ".join(str(x) for x in v))
    r.destroy()
