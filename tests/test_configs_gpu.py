"""BASELINE.json configs at their FULL sizes, driver-run (-m gpu): every config the bench does not
time is rendered once by the HIP kernel through the C ABI and checked against the CPU oracle
where the oracle finishes in seconds (whole rows, tile boundaries, whole small frames), and through
size-independent properties on the rest of the frame.  Bar: bit equality on all 14 channels.

  config 2: 1024 x 1024 x 1024 spp, 9-sphere Cornell box  -> the WHOLE frame vs the oracle, both generators
  config 3: 4096 x 4096 x 64 spp, row tiles of 512 rows    -> rows 0 / 2047 / 4095 + the 511|512 tile boundary
  config 4: 1000 random spheres, 1024 x 1024 x 256 spp     -> four full-width rows, closed and open scene
  config 5: 512 x 512 x 4 spp per frame, 8 bounces         -> three consecutive whole frames, persisted generator
"""
import numpy as np
import pytest

from test_parity_gpu import assert_bit_exact

pytestmark = pytest.mark.gpu


def _frame_properties(full, closed=True):
    assert np.isfinite(full).all()
    assert (full[..., 10:] >= 0).all()                                   # variances
    assert (full[..., 6:9] >= 0).all() and (full[..., 6:9] <= 1.0).all()  # albedo = mean of colours in [0,1]
    nrm = np.linalg.norm(full[..., 3:6], axis=-1)
    assert (nrm <= 1.0 + 1e-3).all()                                      # mean of unit normals
    if closed:
        assert (full[..., 9] > 0).all()                                   # every primary ray hits a wall


# ---- config 3 -------------------------------------------------------------------------------------------
def test_config3_4096_frame_rows_and_tile_boundary(pt, oracle, gpu):
    """The 8-GPU configuration's frame on one GPU: offsets up to 2.3e8 floats, pixel ids up to 1.7e7.
    (1) whole frame rendered once, property-checked; (2) rows 0, 2047, 4095 and the two rows either side
    of the first 512-row tile boundary equal the oracle; (3) rank 1's tile (rows 512..1023), rendered on
    its own exactly as an 8-rank run does, equals the same rows of the full frame."""
    size, spp = 4096, 64
    basis = pt.camera_basis(width=size, height=size)
    scene = pt.scene_cornell()
    r = pt.Renderer(size, size, spp)  # persisted xorwow state: 403 MB, like Renderer::d_states
    d_scene, n = pt.upload_scene(scene)
    d_out = pt.DeviceBuffer(size * size * 56)
    ms = r.render(d_out.ptr, d_scene.ptr, n, basis)
    full = d_out.download(np.float32, (size, size, 14))
    r.destroy()
    d_out.free()
    _frame_properties(full)
    for row in (0, 511, 512, 2047, 4095):
        ref = oracle.render(size, size, spp, spheres=scene, basis=basis, row_begin=row, row_end=row + 1)
        assert_bit_exact(full[row:row + 1], ref, f"config 3 full frame row {row}")
    tile, tile_ms = pt.render_frame(size, size, spp, basis=basis, row_begin=512, row_end=1024)
    assert_bit_exact(tile, full[512:1024], "config 3 tile rows 512..1023 vs the full frame")
    print(f"config 3: full frame {ms:.1f} ms ({size * size * spp / ms / 1e3:.0f} Msamples/s), 512-row tile {tile_ms:.2f} ms")


# ---- config 4 -------------------------------------------------------------------------------------------
@pytest.mark.parametrize("with_walls", [True, False], ids=["closed", "open"])
def test_config4_full_size_rows(pt, oracle, gpu, with_walls):
    """1000 random spheres at 1024 x 1024 x 256 spp with the automatic variant (uniform grid): four
    full-width rows of the frame against the oracle (1 Msample x 1007 spheres x <= 5 bounces), the same
    rows rendered as one-row tiles, and frame properties."""
    size, spp = 1024, 256
    scene = pt.scene_random(1000, seed=1, with_walls=with_walls)
    basis = pt.camera_basis(width=size, height=size)
    r = pt.Renderer(size, size, spp)
    d_scene, n = pt.upload_scene(scene)
    assert r.kernel_info(n)["variant"] == 13
    d_out = pt.DeviceBuffer(size * size * 56)
    ms = r.render(d_out.ptr, d_scene.ptr, n, basis)
    full = d_out.download(np.float32, (size, size, 14))
    r.destroy()
    _frame_properties(full, closed=with_walls)
    for row in (0, 300, 700, 1023):
        ref = oracle.render(size, size, spp, spheres=scene, basis=basis, row_begin=row, row_end=row + 1)
        assert_bit_exact(full[row:row + 1], ref, f"config 4 walls={with_walls} row {row}")
    tile, _ = pt.render_frame(size, size, spp, spheres=scene, basis=basis, row_begin=300, row_end=301)
    assert_bit_exact(tile, full[300:301], "config 4 one-row tile vs the full frame")
    print(f"config 4 walls={with_walls}: {ms:.1f} ms, {size * size * spp / ms / 1e3:.0f} Msamples/s")


# ---- config 5 -------------------------------------------------------------------------------------------
@pytest.mark.parametrize("rng", [0, 1], ids=["xorwow", "philox"])
def test_config5_whole_frames_with_persisted_state(pt, oracle, gpu, rng):
    """The interactive shape at full size: 512 x 512, 4 spp per frame, 8-bounce cap, three consecutive
    frames into the same device buffer; the xorwow state carries over (pathtrace.cu:212,256), philox is
    keyed by the frame counter.  Whole frames against the oracle."""
    size, spp, mb = 512, 4, 8
    basis = pt.camera_basis(width=size, height=size)
    scene = pt.scene_cornell()
    r = pt.Renderer(size, size, spp, max_bounces=mb, rng_mode=rng)
    d_scene, n = pt.upload_scene(scene)
    d_out = pt.DeviceBuffer(size * size * 56)
    st = oracle.setup_random(size, size) if rng == 0 else None
    for frame in range(3):
        r.render(d_out.ptr, d_scene.ptr, n, basis)
        ref = oracle.render(size, size, spp, spheres=scene, basis=basis, max_bounces=mb, rng_mode=rng, rng_state=st, frame=frame)
        assert_bit_exact(d_out.download(np.float32, (size, size, 14)), ref, f"config 5 rng {rng} frame {frame}")
    if rng == 0:
        assert np.array_equal(r.get_rng_state(), st)
    r.destroy()


# ---- config 2, whole frame ---------------------------------------------------------------------------------
@pytest.mark.parametrize("rng", [0, 1], ids=["xorwow", "philox"])
def test_config2_whole_frame_against_the_oracle(pt, oracle, gpu, rng):
    """The headline frame itself -- 1.07e9 samples, 4.8e10 sphere tests -- against the CPU oracle on all
    usable host cores (about 30 s on the GPU box's 16-core share): all 14 680 064 floats bit for bit."""
    size, spp = 1024, 1024
    basis = pt.camera_basis(width=size, height=size)
    img, ms = pt.render_frame(size, size, spp, basis=basis, rng_mode=rng)
    ref = oracle.render(size, size, spp, spheres=pt.scene_cornell(), basis=basis, rng_mode=rng)
    assert_bit_exact(img, ref, f"config 2 whole frame rng {rng}")
    print(f"config 2 rng {rng}: {ms:.1f} ms")


# ---- edge: empty scene with a NULL sphere pointer, every kernel family ---------------------------------------
def test_empty_scene_null_pointer_every_family(pt, gpu):
    """n_spheres == 0 with d_spheres == NULL is accepted by the ABI; no kernel family may touch sphere 0."""
    basis = pt.camera_basis(width=32, height=32)
    d_out = pt.DeviceBuffer(32 * 32 * 56)
    for v in (0, 6, 8, 10, 13, None):
        r = pt.Renderer(32, 32, 3, variant=v)
        pt.check(pt.lib.pt_memset(d_out.ptr, 0xFF, 32 * 32 * 56))
        r.render(d_out.ptr, None, 0, basis)
        assert np.all(d_out.download(np.float32, (32, 32, 14)) == 0), f"variant {v}"
        r.destroy()
