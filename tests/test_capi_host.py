"""CPU tests of the C ABI: the library loads, exports every symbol include/ptcore.h declares,
validates arguments, produces the host-side inputs (scene tables, camera basis) identically to
the oracle, and FAILS LOUDLY (no fallback) when asked to compute without a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def declared_symbols(header="ptcore.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pt_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(pt):
    names = declared_symbols()
    assert len(names) >= 24
    raw = ctypes.CDLL(pt.LIB_PATH)
    missing = [n for n in names if not hasattr(raw, n)]
    assert not missing, missing
    # and the ctypes table used by tests/bench covers the whole header
    assert sorted(pt.ABI) == names


def test_product_library_is_lean_and_the_lab_library_is_a_superset(pt, lab):
    """libptcore.so ships no diagnostics and no experimental kernels; libptcore_lab.so = the same ABI + include/ptcore_lab.h."""
    import subprocess

    prod = subprocess.run(["nm", "-D", "--defined-only", pt.LIB_PATH], capture_output=True, text=True).stdout
    assert "pt_debug_" not in prod and not pt.IS_LAB
    lab_names = declared_symbols("ptcore_lab.h")
    assert lab_names == sorted(lab.LAB_ABI) and all(n.startswith("pt_debug_") for n in lab_names)
    raw = ctypes.CDLL(lab.LIB_PATH)
    assert lab.IS_LAB and not [n for n in declared_symbols() + lab_names if not hasattr(raw, n)]
    assert lab.build_fingerprint() == pt.build_fingerprint() + "-lab"  # built from the same sources
    assert os.path.getsize(pt.LIB_PATH) < os.path.getsize(lab.LIB_PATH)


def test_abi_version_and_struct_layout(pt):
    assert pt.lib.pt_abi_version() == 6
    assert pt.SPHERE_DTYPE.itemsize == 40          # include/Scene.h:7-14
    assert ctypes.sizeof(pt.RendererOpts) == 48 and ctypes.sizeof(pt.MgpuOpts) == 16
    o = pt.RendererOpts()
    pt.lib.pt_renderer_opts_default(ctypes.byref(o))
    assert (o.max_bounces, o.rng_mode, o.seed, o.persist_rng, o.variant, o.layout) == (5, 0, 0, 1, -1, pt.LAYOUT_INTERLEAVED)


def test_cornell_scene_matches_oracle_table(pt, oracle):
    assert pt.scene_cornell().tobytes() == oracle.scene_cornell().tobytes()
    s = pt.scene_cornell()
    assert s["radius"][8] == 600.0 and tuple(s["emission"][8]) == (4.0, np.float32(3.6), np.float32(3.2))


def test_random_scene_is_seeded_and_bounded(pt):
    a, b, c = pt.scene_random(1000, seed=7), pt.scene_random(1000, seed=7), pt.scene_random(1000, seed=8)
    assert a.tobytes() == b.tobytes() and a.tobytes() != c.tobytes()
    assert a[:6].tobytes() == pt.scene_cornell()[:6].tobytes() and a[6].tobytes() == pt.scene_cornell()[8].tobytes()
    r = a[7:]
    assert (r["radius"] >= 0.5).all() and (r["radius"] < 3.0).all()
    assert (r["pos"][:, 0] >= 1).all() and (r["pos"][:, 0] < 99).all()
    assert (r["pos"][:, 2] >= 0).all() and (r["pos"][:, 2] < 170).all()
    assert 0 < (r["emission"][:, 0] > 0).sum() < 40
    op = pt.scene_random(50, seed=7, with_walls=False)
    assert (op["radius"] < 3.0).all()
    with pytest.raises(pt.PtError):
        pt.scene_random(3, seed=1, with_walls=True)


@pytest.mark.parametrize("pose", [((50.0, 52.0, 295.6), -90.0, 0.0), ((10.0, 20.0, 30.0), -45.0, 10.0),
                                  ((50.0, 52.0, 100.0), -120.0, -25.0)])
def test_camera_basis_matches_oracle_bitwise(pt, oracle, pose):
    pos, yaw, pitch = pose
    for w, h in ((256, 256), (1024, 1024)):
        a = pt.camera_basis(pos, yaw, pitch, w, h)
        b = oracle.camera_basis(pos, yaw, pitch, w, h)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_camera_basis_with_explicit_world_up(pt, oracle):
    """The scalar Camera constructor's WorldUp (Camera.h:63-70): (0,1,0) is the vector constructor's basis; a tilted
    or inverted up vector matches the oracle's restatement bit for bit and really changes the basis."""
    pos = (50.0, 52.0, 295.6)
    base = pt.camera_basis(pos, -90.0, 0.0, 512, 512)
    assert np.array_equal(pt.camera_basis(pos, -90.0, 0.0, 512, 512, world_up=(0, 1, 0)).view(np.uint32), base.view(np.uint32))
    for up in ((0.0, -1.0, 0.0), (0.3, 1.0, 0.1), (1.0, 0.0, 0.0)):
        a = pt.camera_basis(pos, -75.0, 5.0, 640, 480, world_up=up)
        b = oracle.camera_basis(pos, -75.0, 5.0, 640, 480, world_up=up)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), up
    flipped = pt.camera_basis(pos, -90.0, 0.0, 512, 512, world_up=(0, -1, 0)).reshape(4, 3)
    assert np.allclose(flipped[0], base.reshape(4, 3)[3], atol=1e-6)  # upside down: corner (-1,-1) <-> (+1,+1)


def test_argument_validation(pt):
    for bad in (dict(width=0), dict(spp=0), dict(max_bounces=-1), dict(rng_mode=7), dict(row_begin=5, row_end=3),
                dict(row_begin=0, row_end=99), dict(variant=999)):
        kw = dict(width=16, height=16, spp=1)
        kw.update(bad)
        w, h, spp = kw.pop("width"), kw.pop("height"), kw.pop("spp")
        with pytest.raises(pt.PtError) as e:
            pt.Renderer(w, h, spp, **kw)
        assert e.value.code == -1, (bad, str(e.value))


def test_compute_fails_loudly_without_gpu(pt):
    """No CPU fallback: on a machine without a HIP device every compute entry point errors."""
    try:
        n = pt.device_count()
    except pt.PtError as e:
        n = 0
        assert e.code in (-2, -3)
    if n > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(pt.PtError) as e:
        pt.Renderer(16, 16, 1)
    assert e.value.code in (-2, -3) and "device" in str(e.value).lower()
    with pytest.raises(pt.PtError):
        pt.DeviceBuffer(1024)


def test_mgpu_fails_loudly_without_gpu_and_validates_arguments(pt):
    """The multi-GPU entry has no fallback either, and rejects nonsense before touching a device."""
    with pytest.raises(pt.PtError) as e:
        pt.MultiRenderer([], 16, 16, 1)
    assert e.value.code == -1
    with pytest.raises(pt.PtError) as e:
        pt.MultiRenderer([0], 0, 16, 1)
    assert e.value.code == -1
    try:
        n = pt.device_count()
    except pt.PtError:
        n = 0
    if n == 0:
        with pytest.raises(pt.PtError) as e:
            pt.MultiRenderer([0, 1], 16, 16, 1)
        assert e.value.code == -3 and "device" in str(e.value).lower()
    mo = pt.MgpuOpts()
    pt.lib.pt_mgpu_opts_default(ctypes.byref(mo))
    assert (mo.gather, mo.timeout_ms) == (pt.GATHER_AUTO, 60000)


def test_reference_construction_sequence_compiles_against_the_look_alikes(pt, tmp_path):
    """src/main.cu:125-131,182-183 verbatim -- Camera(glm::vec3(...), yaw, pitch) included -- builds with plain g++
    against cuda-pathtrace_amd/host/*.h; without a GPU it stops at the first device call with the GPUassert line."""
    import subprocess

    exe = str(tmp_path / "main_cu_lines")
    libdir = os.path.join(ROOT, "cuda-pathtrace_amd")
    res = subprocess.run(["g++", "-std=c++11", "-Wall", "-Wextra", "-Werror", "-I", ROOT, "-include", "iostream", "-o", exe,
                          os.path.join(ROOT, "tests", "cpp", "main_cu_camera_line.cpp"), "-L", libdir, "-lptcore",
                          "-Wl,-rpath," + libdir], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    run = subprocess.run([exe], capture_output=True, text=True)
    try:
        have_gpu = pt.device_count() > 0
    except pt.PtError:
        have_gpu = False
    if have_gpu:
        assert run.returncode == 0 and "Render completed in" in run.stdout
    else:
        assert run.returncode != 0 and "GPUassert:" in run.stderr


@pytest.mark.gpu
def test_reference_construction_sequence_runs_on_the_gpu(pt, gpu, tmp_path):
    test_reference_construction_sequence_compiles_against_the_look_alikes(pt, tmp_path)


def _build_abi_example(tmp_path):
    import subprocess

    exe = str(tmp_path / "abi_example")
    libdir = os.path.join(ROOT, "cuda-pathtrace_amd")
    res = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", ROOT, "-o", exe,
                          os.path.join(ROOT, "tests", "cpp", "abi_example.c"), "-L", libdir, "-lptcore",
                          "-Wl,-rpath," + libdir], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    return exe


def test_header_is_strict_c99_and_example_fails_loudly_without_gpu(pt, tmp_path):
    """INTEGRATION.md's raw C binding compiled with gcc -std=c99 -pedantic against include/ptcore.h."""
    import subprocess

    run = subprocess.run([_build_abi_example(tmp_path)], capture_output=True, text=True)
    try:
        have_gpu = pt.device_count() > 0
    except pt.PtError:
        have_gpu = False
    if have_gpu:
        assert run.returncode == 0 and run.stdout.startswith("ok ")
    else:
        assert run.returncode == 3 and "device" in run.stderr.lower()


@pytest.mark.gpu
def test_abi_example_success_path_on_the_gpu(pt, oracle, gpu, tmp_path):
    """The same plain-C program on a real device: renders 64 x 64 x 4 spp through the raw C ABI, prints the
    kernel time, and -- with a file name -- dumps the frame, which must equal the oracle bit for bit."""
    import subprocess

    dump = str(tmp_path / "frame.f32")
    run = subprocess.run([_build_abi_example(tmp_path), dump], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0, run.stderr
    assert run.stdout.startswith("ok ") and "abi %d" % pt.lib.pt_abi_version() in run.stdout
    img = np.fromfile(dump, dtype=np.float32).reshape(64, 64, 14)
    ref = oracle.render(64, 64, 4, spheres=pt.scene_cornell(), basis=pt.camera_basis(width=64, height=64))
    assert np.array_equal(img.view(np.uint32), ref.view(np.uint32))
