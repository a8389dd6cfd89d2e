import os, sys
sys.path.insert(0, os.getcwd())
name = sys.argv[1]
if name != "main":
    os.environ["PT_LIB_OVERRIDE"] = os.path.join(os.getcwd(), "cuda-pathtrace_amd", "alt", name, "libptcore.so")
import __graft_entry__ as ge
pt = ge.load_package(); pt.set_device(0)
basis = pt.camera_basis(width=1024, height=1024)
d_scene, n = pt.upload_scene(pt.scene_cornell())
d_out = pt.DeviceBuffer(1024 * 1024 * 56)
out = []
for rows in (1024, 512, 256, 128):
    for rng in (0, 1):
        r = pt.Renderer(1024, 1024, 1024, rng_mode=rng, row_begin=0, row_end=rows)
        ms = min(r.render(d_out.ptr, d_scene.ptr, n, basis) for _ in range(3))
        out.append(f"{rows}/{'px'[rng]}/v{r.kernel_info(n)['variant']} {ms:.3f}")
        r.destroy()
r = pt.Renderer(1024, 1024, 1024, fast_math=True)
out.append(f"fast {min(r.render(d_out.ptr, d_scene.ptr, n, basis) for _ in range(3)):.3f}")
print(name, " | ".join(out), flush=True)
