import os, subprocess, sys, json
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 2 and sys.argv[1] == "--child":
    sys.path.insert(0, root)
    os.environ["PT_LIB_OVERRIDE"] = sys.argv[2]
    import __graft_entry__ as ge
    pt = ge.load_package(); pt.set_device(0)
    basis = pt.camera_basis(width=1024, height=1024)
    d_out = pt.DeviceBuffer(1024*1024*56)
    res = []
    for n, walls in ((1000, True), (1000, False), (300, True), (100, True)):
        sc = pt.scene_random(n, seed=1, with_walls=walls)
        d_scene, ns = pt.upload_scene(sc)
        h = pt.grid_header(sc)
        r = pt.Renderer(1024, 1024, 8, variant=11)
        ms = min(r.render(d_out.ptr, d_scene.ptr, ns, basis) for _ in range(2))
        r.destroy()
        res.append(f"n={n} walls={int(walls)} dims={h['dims']} items={h['n_items']} {ms:.2f} ms")
    print(os.path.basename(os.path.dirname(sys.argv[2])), " | ".join(res), flush=True)
else:
    for name in sys.argv[1:]:
        lib = os.path.join(root, "cuda-pathtrace_amd", "alt", name, "libptcore.so") if name != "main" else os.path.join(root, "cuda-pathtrace_amd", "libptcore.so")
        subprocess.call([sys.executable, __file__, "--child", lib])
