"""pt_mgpu_* (include/ptcore.h): one frame row-tiled over several devices by ONE process -- one host thread
per device inside libptcore, grouped RCCL send/recv (or peer copies) into the root's frame.  The driver's
test box has one GPU, so the multi-rank machinery is exercised with ranks SHARING device 0 (peer-copy
exchange; RCCL refuses duplicate devices in one communicator) and with a one-rank RCCL communicator whose
tile is forced through the exchange (self send/recv: ncclCommInitAll, ncclGroupStart/End, ncclSend, ncclRecv
all really run).  With two or more GPUs visible the real thing runs too.  Frames must equal the
single-renderer frame and the oracle bit for bit: tiling and exchange may not change a pixel."""
import os
import subprocess

import numpy as np
import pytest

from test_parity_gpu import assert_bit_exact

pytestmark = pytest.mark.gpu


def _render_mgpu(pt, devices, size, spp, frames=1, scene=None, **kw):
    scene = pt.scene_cornell() if scene is None else scene
    basis = pt.camera_basis(width=size, height=size)
    m = pt.MultiRenderer(devices, size, size, spp, **kw)
    d_scene, n = pt.upload_scene(scene)
    d_out = pt.DeviceBuffer(size * size * 56)
    pt.check(pt.lib.pt_memset(d_out.ptr, 0xFF, size * size * 56))
    pt.check(pt.lib.pt_device_synchronize())
    out = []
    for _ in range(frames):
        ms = m.render(d_out.ptr, d_scene.ptr, n, basis)
        assert ms > 0
        out.append(d_out.download(np.float32, (size, size, 14)))
    info = {"backend": m.backend(), "tiles": [m.tile(r) for r in range(len(devices))]}
    m.destroy()
    return out, info


@pytest.mark.parametrize("rng", [0, 1], ids=["xorwow", "philox"])
def test_ranks_sharing_one_device_peer_copy_exchange(pt, oracle, gpu, rng):
    """3 ranks (ragged: 100 rows -> 34/33/33) on device 0, every tile rendered into its own buffer and moved
    into the frame by the exchange step; two frames (persisted generator state per tile)."""
    size, spp = 100, 4
    frames, info = _render_mgpu(pt, [0, 0, 0], size, spp, frames=2, rng_mode=rng, force_exchange=True)
    assert info["backend"] == "hipMemcpyPeerAsync"
    assert [t["rows"] for t in info["tiles"]] == [(0, 34), (34, 67), (67, 100)]
    assert all(t["kernel_ms"] > 0 for t in info["tiles"])
    basis = pt.camera_basis(width=size, height=size)
    st = oracle.setup_random(size, size) if rng == 0 else None
    for f, img in enumerate(frames):
        ref = oracle.render(size, size, spp, spheres=pt.scene_cornell(), basis=basis, rng_mode=rng, rng_state=st, frame=f)
        assert_bit_exact(img, ref, f"3 ranks on one device, frame {f}")


def test_in_place_tiles_without_exchange(pt, oracle, gpu):
    """Ranks on the root's device render straight into the frame (no exchange needed): 5 ranks, 64 rows."""
    frames, info = _render_mgpu(pt, [0] * 5, 64, 3)
    ref = oracle.render(64, 64, 3, spheres=pt.scene_cornell(), basis=pt.camera_basis(width=64, height=64))
    assert_bit_exact(frames[0], ref, "5 in-place tiles")


def test_more_ranks_than_rows(pt, oracle, gpu):
    frames, info = _render_mgpu(pt, [0] * 6, 4, 2, force_exchange=True)
    assert [t["rows"][1] - t["rows"][0] for t in info["tiles"]] == [1, 1, 1, 1, 0, 0]
    ref = oracle.render(4, 4, 2, spheres=pt.scene_cornell(), basis=pt.camera_basis(width=4, height=4))
    assert_bit_exact(frames[0], ref, "6 ranks, 4 rows")


def test_rccl_exchange_on_one_gpu_self_send_recv(pt, oracle, gpu):
    """PT_FORCE_MGPU semantics: a one-rank RCCL communicator; the tile is rendered into the tile buffer and
    placed in the frame by ncclSend/ncclRecv to self inside one group.  1000-sphere scene as well (the scene
    replica path is not taken on the root, the grid kernel is)."""
    size, spp = 96, 4
    frames, info = _render_mgpu(pt, [0], size, spp, frames=2, force_exchange=True, gather=pt.GATHER_RCCL, timeout_ms=30000)
    assert info["backend"].startswith("rccl")
    basis = pt.camera_basis(width=size, height=size)
    st = oracle.setup_random(size, size)
    for f, img in enumerate(frames):
        ref = oracle.render(size, size, spp, spheres=pt.scene_cornell(), basis=basis, rng_state=st)
        assert_bit_exact(img, ref, f"rccl self exchange frame {f}")
    scene = pt.scene_random(400, seed=5)
    frames, _ = _render_mgpu(pt, [0], 48, 2, scene=scene, force_exchange=True, gather=pt.GATHER_RCCL, timeout_ms=30000)
    assert_bit_exact(frames[0], oracle.render(48, 48, 2, spheres=scene, basis=pt.camera_basis(width=48, height=48)), "rccl, 400 spheres")


@pytest.mark.parametrize("rng", [0, 1], ids=["xorwow", "philox"])
@pytest.mark.parametrize("backend", ["copy", "rccl"])
def test_banded_tiles_pipeline_the_exchange_inside_a_frame(pt, oracle, gpu, rng, backend):
    """Round 4: a rank renders its tile as row bands and band b's part of the exchange travels while band b + 1 renders
    (include/ptcore.h, pt_mgpu_opts.bands).  Ragged bands (3 ranks x 3 bands over 100 rows: 34 -> 12/11/11, 33 -> 11/11/11),
    more bands than a tile has rows, two frames with the generator state of every band carried over, both generators, the
    peer-copy exchange between ranks sharing the device and the RCCL exchange (one-rank communicator, self send/recv per band):
    the frame may not change by a bit."""
    size, spp = 100, 4
    basis = pt.camera_basis(width=size, height=size)
    if backend == "copy":
        cases = [([0, 0, 0], dict(bands=3, force_exchange=True)), ([0, 0], dict(bands=64, force_exchange=True)), ([0, 0, 0], dict(bands=5))]
    else:
        cases = [([0], dict(bands=4, force_exchange=True, gather=pt.GATHER_RCCL, timeout_ms=30000))]
    for devices, kw in cases:
        frames, info = _render_mgpu(pt, devices, size, spp, frames=2, rng_mode=rng, **kw)
        assert info["backend"].startswith("rccl" if backend == "rccl" else "hipMemcpyPeerAsync")
        st = oracle.setup_random(size, size) if rng == 0 else None
        for f, img in enumerate(frames):
            ref = oracle.render(size, size, spp, spheres=pt.scene_cornell(), basis=basis, rng_mode=rng, rng_state=st, frame=f)
            assert_bit_exact(img, ref, f"{len(devices)} ranks, {kw}, frame {f}")


def test_band_policy_and_frame_stats(pt, gpu):
    """Automatic bands: none unless a tile crosses a link (ranks sharing the root's device exchange at HBM speed); between distinct
    devices, bands of at least eight one-lane waves per SIMD (BASELINE configs[2]: 512 x 4096 -> 4 bands; configs[1]'s tiles of 2
    waves per SIMD: 1).  pt_mgpu_frame_stats reports them with the render / exposed-exchange split of the last frame."""
    m = pt.MultiRenderer([0], 64, 64, 1)
    assert m.frame_stats()["bands"] == 1
    m.destroy()
    m = pt.MultiRenderer([0], 4096, 512, 1, force_exchange=True)
    assert m.frame_stats()["bands"] == 1
    m.destroy()
    if pt.device_count() >= 2:
        m = pt.MultiRenderer([0, 1], 4096, 1024, 1)
        assert m.frame_stats()["bands"] == 4
        m.destroy()
        m = pt.MultiRenderer([0, 1], 1024, 256, 1)
        assert m.frame_stats()["bands"] == 1
        m.destroy()
    m = pt.MultiRenderer([0], 4096, 512, 1, force_exchange=True, bands=4)
    assert m.frame_stats()["bands"] == 4
    d_scene, n = pt.upload_scene(pt.scene_cornell())
    d_out = pt.DeviceBuffer(4096 * 512 * 56)
    ms = m.render(d_out.ptr, d_scene.ptr, n, pt.camera_basis(width=4096, height=512))
    fs = m.frame_stats()
    assert fs["render_ms"] > 0 and 0 <= fs["exposed_ms"] <= ms and abs(fs["render_ms"] + fs["exposed_ms"] - ms) < 1e-3 * ms + 1e-3
    m.destroy()
    with pytest.raises(pt.PtError):
        pt.MultiRenderer([0], 16, 16, 1, bands=65)


def test_real_multi_gpu_rccl_gather(pt, oracle, gpu):
    """Distinct devices, RCCL over xGMI: only where the box has them (the driver's test box has one GPU)."""
    n = pt.device_count()
    if n < 2:
        pytest.skip("one GPU visible")
    n = min(n, 8)
    size, spp = 256, 8
    frames, info = _render_mgpu(pt, list(range(n)), size, spp, frames=2)
    assert info["backend"].startswith("rccl")
    basis = pt.camera_basis(width=size, height=size)
    st = oracle.setup_random(size, size)
    for f, img in enumerate(frames):
        ref = oracle.render(size, size, spp, spheres=pt.scene_cornell(), basis=basis, rng_state=st)
        assert_bit_exact(img, ref, f"{n} GPUs frame {f}")


def test_argument_errors(pt, gpu):
    with pytest.raises(pt.PtError) as e:
        pt.MultiRenderer([0, 99], 16, 16, 1)
    assert e.value.code == -1 and "device 99" in str(e.value)
    with pytest.raises(pt.PtError) as e:
        pt.MultiRenderer([0, 0], 16, 16, 1, gather=pt.GATHER_RCCL)
    assert e.value.code == -1 and "distinct" in str(e.value)
    with pytest.raises(pt.PtError):
        pt.MultiRenderer([], 16, 16, 1)


def test_cli_gpus_flag_forced_exchange(pt, oracle, gpu, tmp_path):
    """pathtrace --gpus 1 with PT_FORCE_MGPU=1: MultiRenderer + RCCL exchange behind the reference's CLI;
    the saved EXR must hold the oracle's frame."""
    from conftest import ROOT
    from test_parity_gpu import _read_feature_exr

    exe = os.path.join(ROOT, "cuda-pathtrace_amd", "pathtrace")
    out = str(tmp_path / "tiled")
    env = dict(os.environ, PT_FORCE_MGPU="1", PT_MGPU_TIMEOUT_MS="30000")
    res = subprocess.run([exe, "--size", "64", "-s", "4", "--gpus", "1", "--nobitmap", "-o", out], capture_output=True, text=True,
                         timeout=180, env=env)
    assert res.returncode == 0, res.stderr
    assert "Row-tiled over 1 device(s)" in res.stdout and "Exchange: rccl" in res.stdout and "Tile kernel times:" in res.stdout
    names, planes = _read_feature_exr(out + ".exr")
    ref = oracle.render(64, 64, 4, spheres=pt.scene_cornell(), basis=pt.camera_basis(width=64, height=64))
    for nm, ch in (("Color.R", 0), ("Normal.Y", 4), ("DepthVar.Z", 13)):
        assert np.array_equal(planes[names.index(nm)].view(np.uint32), ref[..., ch].view(np.uint32)), nm
    bad = subprocess.run([exe, "--size", "16", "-s", "1", "--gpus", "64", "-o", out], capture_output=True, text=True, timeout=60)
    assert bad.returncode != 0 and "GPUassert:" in bad.stderr


# ---- failure paths (lab library: PT_LAB_MGPU_STALL holds one rank's stream past the frame's deadline) --------------------
def _stalled(lab, devices, monkeypatch, rank, **kw):
    monkeypatch.setenv("PT_LAB_MGPU_STALL", f"{rank}:1500")
    size = 48
    m = lab.MultiRenderer(devices, size, size, 2, timeout_ms=200, **kw)
    d_scene, n = lab.upload_scene(lab.scene_cornell())
    d_out = lab.DeviceBuffer(size * size * 56)
    return m, d_scene, n, d_out, lab.camera_basis(width=size, height=size)


def test_timeout_peer_copy_ranks_then_object_is_dead(lab, gpu, monkeypatch):
    """A rank whose stream does not drain before the deadline: PT_ETIMEOUT (not a hang, not PT_OK), every later render is
    refused with PT_ECOMM, and destroy returns.  3 ranks sharing device 0, peer-copy exchange; rank 1 stalls for 1.5 s
    against a 200 ms deadline."""
    import time
    m, d_scene, n, d_out, basis = _stalled(lab, [0, 0, 0], monkeypatch, 1, force_exchange=True)
    t0 = time.perf_counter()
    with pytest.raises(lab.PtError) as e:
        m.render(d_out.ptr, d_scene.ptr, n, basis)
    assert e.value.code == -7 and "rank 1" in str(e.value) and time.perf_counter() - t0 < 1.4  # returned at the deadline, not after the stall
    monkeypatch.delenv("PT_LAB_MGPU_STALL")
    with pytest.raises(lab.PtError) as e:
        m.render(d_out.ptr, d_scene.ptr, n, basis)
    assert e.value.code == -6 and "destroy" in str(e.value)
    m.destroy()  # joins the workers; the stalled stream drains on its own
    lab.check(lab.lib.pt_device_synchronize())
    # a fresh object on the same device works
    m2 = lab.MultiRenderer([0, 0, 0], 48, 48, 2, force_exchange=True)
    assert m2.render(d_out.ptr, d_scene.ptr, n, basis) > 0
    m2.destroy()


def test_timeout_rccl_communicator_is_aborted(lab, gpu, monkeypatch):
    """The same with the RCCL exchange (one-rank communicator, forced self send/recv): the deadline passes while the grouped
    send/recv is still queued behind the stalled stream, the rank aborts its communicator (ncclCommAbort) and reports
    PT_ETIMEOUT; the object refuses further frames and can be destroyed."""
    m, d_scene, n, d_out, basis = _stalled(lab, [0], monkeypatch, 0, force_exchange=True, gather=lab.GATHER_RCCL)
    assert m.backend().startswith("rccl")
    with pytest.raises(lab.PtError) as e:
        m.render(d_out.ptr, d_scene.ptr, n, basis)
    assert e.value.code == -7 and "aborted" in str(e.value)
    monkeypatch.delenv("PT_LAB_MGPU_STALL")
    with pytest.raises(lab.PtError) as e:
        m.render(d_out.ptr, d_scene.ptr, n, basis)
    assert e.value.code == -6
    m.destroy()
    lab.check(lab.lib.pt_device_synchronize())


def test_render_failure_of_one_rank_does_not_stall_the_others(pt, gpu):
    """A rank that cannot render (here: every rank, a scene beyond the variant's staging limit is not reachable in the product
    library, so a NULL scene with a positive count) fails the call at once with that rank's error -- not after the deadline --
    and the object stays usable."""
    import time
    size = 32
    m = pt.MultiRenderer([0, 0], size, size, 2, force_exchange=True, timeout_ms=5000)
    d_scene, n = pt.upload_scene(pt.scene_cornell())
    d_out = pt.DeviceBuffer(size * size * 56)
    basis = pt.camera_basis(width=size, height=size)
    t0 = time.perf_counter()
    with pytest.raises(pt.PtError) as e:
        m.render(d_out.ptr, None, 9, basis)
    assert e.value.code == -1 and time.perf_counter() - t0 < 2.0
    assert m.render(d_out.ptr, d_scene.ptr, n, basis) > 0  # argument errors leave the object usable
    m.destroy()
