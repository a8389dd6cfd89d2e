"""Sample chunking must never produce a silently wrong frame: a workgroup whose predecessor never signals raises the renderer's
device error word and leaves its pixel block untouched.  Render() then completes the frame itself (repair launch), an enqueued
frame is reported as PT_EKERNEL; either way the renderer goes on unchunked, bit-exact.
The broken chain is provoked in the lab library (PT_LAB_DEBUG=1: pixel block 0, chunk 0 does not publish its flag)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

PT_EKERNEL = -8


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def test_chunk_switch_and_lazy_buffer(pt, gpu, oracle):
    """opts.chunks: 1 = never, n = n chained workgroups per pixel block, 0 = automatic; the bits do not depend on it."""
    w, h, spp = 40, 24, 601
    basis = pt.camera_basis(width=w, height=h)
    ref = oracle.render(w, h, spp, spheres=pt.scene_cornell(), basis=basis)
    d_scene, n = pt.upload_scene(pt.scene_cornell())
    d_out = pt.DeviceBuffer(w * h * 56)
    blocks = (w * h + 255) // 256
    for chunks, per_block in ((1, 1), (2, 2), (4, 4), (8, 8), (16, 16), (0, 6)):
        r = pt.Renderer(w, h, spp, variant=6, persist_rng=False, chunks=chunks)
        assert r.kernel_info(n)["grid_blocks"] == blocks * per_block, chunks
        r.render(d_out.ptr, d_scene.ptr, n, basis)
        assert np.array_equal(_bits(d_out.download(np.float32, (h, w, 14))), _bits(ref)), f"chunks={chunks}"
        r.destroy()
    # kernels that never chunk report one workgroup per pixel block whatever the option says
    r = pt.Renderer(w, h, spp, variant=10, chunks=8)
    assert r.kernel_info(n)["grid_blocks"] == blocks
    r.destroy()
    r = pt.Renderer(w, h, spp, variant=8, chunks=8, max_bounces=6)  # no reference-configuration build for 6 bounces
    assert r.kernel_info(n)["grid_blocks"] == (w * h * 4 + 255) // 256
    r.destroy()
    with pytest.raises(pt.PtError) as e:
        pt.Renderer(w, h, spp, chunks=17)
    assert e.value.code == -1


@pytest.mark.parametrize("rng", [0, 1], ids=["xorwow", "philox"])
def test_pooled_grid_kernel_chunks(pt, gpu, oracle, rng):
    """Variant 13 chains a pixel's samples through several 512-pixel workgroups too (its frames are few rounds of very long
    workgroups): same bits for every chunk count, ragged sample counts and a ragged last workgroup, two frames with the generator
    state carried over, closed and open scene."""
    w, h, spp = 72, 40, 37
    basis = pt.camera_basis(width=w, height=h)
    blocks = (w * h + 511) // 512
    for walls in (True, False):
        scene = pt.scene_random(300, seed=31, with_walls=walls)
        d_scene, n = pt.upload_scene(scene)
        d_out = pt.DeviceBuffer(w * h * 56)
        for chunks, per_block in ((1, 1), (2, 2), (5, 5), (8, 8), (16, 16)):
            r = pt.Renderer(w, h, spp, variant=13, rng_mode=rng, chunks=chunks)
            assert r.kernel_info(n)["grid_blocks"] == blocks * per_block, chunks
            st = oracle.setup_random(w, h) if rng == 0 else None
            for frame in range(2):
                r.render(d_out.ptr, d_scene.ptr, n, basis)
                ref = oracle.render(w, h, spp, spheres=scene, basis=basis, rng_mode=rng, rng_state=st, frame=frame)
                assert np.array_equal(_bits(d_out.download(np.float32, (h, w, 14))), _bits(ref)), f"walls={walls} chunks={chunks} frame {frame}"
            r.destroy()
    # the automatic policy: a frame of few rounds of workgroups is chunked, a frame of many is not
    r = pt.Renderer(1024, 1024, 256)
    assert r.kernel_info(1000)["variant"] == 13 and r.kernel_info(1000)["grid_blocks"] == 4 * (1024 * 1024 // 512)
    r.destroy()
    r = pt.Renderer(4096, 4096, 64)
    assert r.kernel_info(1000)["grid_blocks"] == 4096 * 4096 // 512  # 32768 workgroups = 64 rounds already
    r.destroy()


@pytest.mark.parametrize("variant", [8, 9], ids=["four_lanes", "two_lanes"])
@pytest.mark.parametrize("rng", [0, 1], ids=["xorwow", "philox"])
def test_split_kernels_chunks(pt, gpu, oracle, rng, variant):
    """Variants 8 and 9 (several lanes per pixel, speculative generator skip-ahead) chain their samples through workgroups as
    well: every lane hands over the features it owns, lane 0 the pixel's TRUE generator state -- also when a speculation
    failed inside the chunk (open scene) and when the sample count is not a multiple of the lanes per pixel or of the chunks."""
    w, h = 40, 24
    basis = pt.camera_basis(width=w, height=h)
    lanes = 4 if variant == 8 else 2
    blocks = (w * h * lanes + 255) // 256
    for name, scene in (("closed", pt.scene_cornell()), ("open", pt.scene_cornell()[[0, 2, 4, 6, 7, 8, 1, 3, 5]][:9])):
        if name == "open":
            scene = scene.copy()
            scene["radius"][6:] = 1.0  # three of the walls become small spheres: paths escape, speculations fail (still 9 spheres: REF build)
        d_scene, n = pt.upload_scene(scene)
        d_out = pt.DeviceBuffer(w * h * 56)
        for spp, chunks in ((37, 2), (37, 5), (64, 4), (13, 3)):
            r = pt.Renderer(w, h, spp, variant=variant, rng_mode=rng, chunks=chunks)
            assert r.kernel_info(n)["grid_blocks"] == blocks * chunks, (spp, chunks)
            st = oracle.setup_random(w, h) if rng == 0 else None
            for frame in range(2):
                r.render(d_out.ptr, d_scene.ptr, n, basis)
                ref = oracle.render(w, h, spp, spheres=scene, basis=basis, rng_mode=rng, rng_state=st, frame=frame)
                assert np.array_equal(_bits(d_out.download(np.float32, (h, w, 14))), _bits(ref)), f"{name} spp {spp} chunks {chunks} frame {frame}"
                if rng == 0:
                    assert np.array_equal(r.get_rng_state(), st), f"{name} spp {spp} chunks {chunks}: generator state after frame {frame}"
            r.destroy()


def _lab_renderers(lab, specs, timeout_ms=150):
    """Renderers of the lab library created with the chunk-flag fault switched on (PT_LAB_DEBUG=1: pixel block 0, chunk 0 never
    publishes its flag) and a short wait limit."""
    old = {k: os.environ.get(k) for k in ("PT_LAB_DEBUG", "PT_CHUNK_TIMEOUT_MS")}
    os.environ["PT_LAB_DEBUG"] = "1"
    os.environ["PT_CHUNK_TIMEOUT_MS"] = str(timeout_ms)
    try:
        return [lab.Renderer(*a, **k) for a, k in specs]
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("variant", [6, 9, 13])
def test_broken_chunk_chain_is_repaired_by_render(lab, gpu, oracle, variant):
    """Synchronous Render(): the chunked launch leaves every pixel block complete or untouched (generator state included), the
    call renders the untouched ones again unchunked, and frame AND persisted generator state are the oracle's -- with the
    reference's default persist_rng, over two frames (the next frame must not continue a corrupted stream)."""
    if variant == 13:
        w, h, spp, chunks = 64, 24, 64, 4
        scene = lab.scene_random(200, seed=5, with_walls=True)
    else:
        w, h, spp, chunks = 64, 16, 640, 4
        scene = lab.scene_cornell()
    basis = lab.camera_basis(width=w, height=h)
    d_scene, n = lab.upload_scene(scene)
    d_out = lab.DeviceBuffer(w * h * 56)
    (r,) = _lab_renderers(lab, [((w, h, spp), dict(variant=variant, chunks=chunks))])
    lanes = {6: 1, 9: 2, 13: 1}[variant]
    block = 512 if variant == 13 else 256
    assert r.kernel_info(n)["grid_blocks"] == chunks * ((w * h * lanes + block - 1) // block)
    st = oracle.setup_random(w, h)
    for frame in range(2):
        d_out.upload(np.full((h, w, 14), -7.0, dtype=np.float32))
        r.render(d_out.ptr, d_scene.ptr, n, basis)  # frame 0: chain broken and repaired; frame 1: unchunked
        ref = oracle.render(w, h, spp, spheres=scene, basis=basis, rng_state=st)
        assert np.array_equal(_bits(d_out.download(np.float32, (h, w, 14))), _bits(ref)), f"frame {frame}"
        assert np.array_equal(r.get_rng_state(), st), f"generator state after frame {frame}"
        assert r.check() == 1  # one repaired frame, nothing pending
        # the renderer has stopped chunking: one workgroup per pixel block
        assert r.kernel_info(n)["grid_blocks"] == (w * h * lanes + block - 1) // block
    r.destroy()


def test_broken_chunk_chain_of_an_enqueued_frame_is_an_error(lab, gpu, oracle):
    """pt_renderer_enqueue cannot repair (nobody waits): the frame is incomplete and pt_renderer_check -- or the next call on the
    renderer -- says so once; the pixel blocks concerned were left untouched, their generator state included, every other block
    is the oracle's; after re-seeding, the renderer (now unchunked) is bit-exact again."""
    w, h, spp = 64, 16, 640
    basis = lab.camera_basis(width=w, height=h)
    ref = oracle.render(w, h, spp, spheres=lab.scene_cornell(), basis=basis)
    d_scene, n = lab.upload_scene(lab.scene_cornell())
    d_out = lab.DeviceBuffer(w * h * 56)
    r, r2 = _lab_renderers(lab, [((w, h, spp), dict(variant=6, chunks=4)), ((w, h, spp), dict(variant=6, chunks=4))])
    fresh = r.get_rng_state()
    d_out.upload(np.full((h, w, 14), -7.0, dtype=np.float32))
    r.enqueue(d_out.ptr, d_scene.ptr, n, basis)
    lab.check(lab.lib.pt_device_synchronize())
    with pytest.raises(lab.PtError) as e:
        r.check()
    assert e.value.code == PT_EKERNEL and "chunk" in str(e.value)
    assert r.check() == 0  # reported once
    got = d_out.download(np.float32, (h, w, 14)).reshape(-1, 14)
    state = r.get_rng_state()
    refp = ref.reshape(-1, 14)
    assert np.all(got[:256] == -7.0) and np.array_equal(state[:256], fresh[:256])  # pixel block 0: untouched
    assert np.array_equal(_bits(got[256:]), _bits(refp[256:]))                     # the others: complete
    r.reset_rng()
    r.enqueue(d_out.ptr, d_scene.ptr, n, basis)  # unchunked from here on
    lab.check(lab.lib.pt_device_synchronize())
    assert r.check() == 0
    assert np.array_equal(_bits(d_out.download(np.float32, (h, w, 14))), _bits(ref))
    r.destroy()

    # without pt_renderer_check the next call on the renderer reports it
    r2.enqueue(d_out.ptr, d_scene.ptr, n, basis)
    lab.check(lab.lib.pt_device_synchronize())
    with pytest.raises(lab.PtError) as e:
        r2.enqueue(d_out.ptr, d_scene.ptr, n, basis)
    assert e.value.code == PT_EKERNEL
    r2.destroy()
