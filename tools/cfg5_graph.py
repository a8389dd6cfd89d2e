#!/usr/bin/env python3
"""BASELINE configs[4] (512 x 512 x 4 spp x 8 bounces, the interactive shape) as a stream of 200 frames: plain enqueues against
a hipGraph replay of the captured frame (torch.cuda.CUDAGraph around pt_renderer_enqueue; xorwow: the generator state lives in
HBM, so a replay IS the next frame).  With and without the fused display pack.  Usage: cfg5_graph.py [frames=200]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
pt = ge.load_package(); pt.set_device(0)
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 200
w = h = 512
dev = torch.device("cuda", 0)
sph = pt.scene_cornell()
d_scene = torch.from_numpy(sph.view("u1").reshape(-1).copy()).to(dev)
frame = torch.empty(h * w * 14, dtype=torch.float32, device=dev)
vtx = torch.empty(h * w * 3, dtype=torch.float32, device=dev)
basis = pt.camera_basis(width=w, height=h)
out = {}
for fused in (False, True):
    r = pt.Renderer(w, h, 4, max_bounces=8)
    r.set_display(vtx.data_ptr() if fused else None)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(5):
            r.enqueue(frame.data_ptr(), d_scene.data_ptr(), len(sph), basis, stream=s.cuda_stream)
        s.synchronize()
        t0 = time.perf_counter()
        for _ in range(frames):
            r.enqueue(frame.data_ptr(), d_scene.data_ptr(), len(sph), basis, stream=s.cuda_stream)
        s.synchronize()
        plain = (time.perf_counter() - t0) / frames * 1e3
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            r.enqueue(frame.data_ptr(), d_scene.data_ptr(), len(sph), basis, stream=s.cuda_stream)
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(frames):
            g.replay()
        torch.cuda.synchronize()
        graph = (time.perf_counter() - t0) / frames * 1e3
    out["fused_display" if fused else "render_only"] = {"enqueue_ms_per_frame": round(plain, 4), "graph_replay_ms_per_frame": round(graph, 4)}
    r.destroy()
print(json.dumps(out))
