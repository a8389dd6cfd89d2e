#!/usr/bin/env python3
"""Row-block load balance on ONE GPU: the kernel time of every contiguous row tile a rank would render at N = 2, 4, 8
(tiling.row_range / pt_mgpu's row_range),
for the closed Cornell box and the 1000-sphere scene with and without walls.  Predicted kernel-part efficiency of a strong-scaling
run = sum of tile times / (N * slowest tile).  Usage: tile_skew.py [reps=2]  -> gpurun_out/tile_skew.json + a table on stdout"""
import json, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import __graft_entry__ as ge
pt = ge.load_package(); pt.set_device(0)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
H = W = 1024
basis = pt.camera_basis(width=W, height=H)
SCENES = (("cfg2 Cornell box 1024 spp", pt.scene_cornell(), 1024), ("cfg4 closed 256 spp", pt.scene_random(1000, seed=1, with_walls=True), 256),
          ("cfg4 open 256 spp", pt.scene_random(1000, seed=1, with_walls=False), 256))
out = {}


def tile_ms(scene, spp, rb, re_):
    r = pt.Renderer(W, H, spp, row_begin=rb, row_end=re_)
    d_scene, n = pt.upload_scene(scene)
    d_out = pt.DeviceBuffer((re_ - rb) * W * 56)
    ms = min(r.render(d_out.ptr, d_scene.ptr, n, basis) for _ in range(reps))
    r.destroy(); d_out.free(); d_scene.free()
    return ms


for name, scene, spp in SCENES:
    full = tile_ms(scene, spp, 0, H)
    rec = {"full_frame_ms": round(full, 3), "contiguous": {}}
    print(f"{name}: full frame {full:.3f} ms")
    for N in (2, 4, 8):
        t = [tile_ms(scene, spp, g * H // N, (g + 1) * H // N) for g in range(N)]
        eff = sum(t) / (N * max(t))
        rec["contiguous"][N] = {"tile_ms": [round(x, 3) for x in t], "efficiency_vs_slowest": round(eff, 4), "speedup_vs_full": round(full / max(t), 3)}
        print(f"  N={N} contiguous tiles ms {[round(x, 2) for x in t]}  balance {eff:.3f}  speed-up {full / max(t):.2f}")
    out[name] = rec
out["fingerprint"] = pt.build_fingerprint()
os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(root, "gpurun_out", "tile_skew.json"), "w"), indent=1)
