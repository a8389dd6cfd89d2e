"""Parity tests proper: the HIP kernel, called through the C ABI, against the CPU oracle on
the same seeded inputs.  Bar: BIT-EXACT on all 14 channels (the kernel implements the
oracle's numeric contract operation for operation), which is far inside north_star's
per-pixel L-inf <= 1e-4."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def assert_bit_exact(img, ref, what):
    assert img.shape == ref.shape
    neq = _bits(img) != _bits(ref)
    if neq.any():
        bad = np.argwhere(neq)
        r, c, ch = bad[0]
        raise AssertionError(
            f"{what}: {neq.sum()} of {neq.size} floats differ ({len(set(map(tuple, bad[:, :2])))} pixels); first at "
            f"[row {r}, col {c}, ch {ch}]: hip {img[r, c, ch]!r} oracle {ref[r, c, ch]!r}; "
            f"max |diff| = {np.nanmax(np.abs(img - ref)):.3g}"
        )


@pytest.mark.parametrize("rng", [0, 1], ids=["xorwow", "philox"])
@pytest.mark.parametrize("size,spp", [(64, 1), (64, 4), (128, 16), (256, 4)])
def test_cornell_bit_exact(pt, oracle, gpu, size, spp, rng):
    basis = pt.camera_basis(width=size, height=size)
    img, ms = pt.render_frame(size, size, spp, basis=basis, rng_mode=rng)
    ref = oracle.render(size, size, spp, spheres=pt.scene_cornell(), basis=basis, rng_mode=rng)
    assert np.isfinite(img).all()
    assert_bit_exact(img, ref, f"cornell {size}x{size}x{spp} rng={rng}")


# ---- committed fixtures: stored vectors, not only the live oracle --------------------------------
@pytest.mark.parametrize("name", ["oracle_64_spp1_xorwow", "oracle_64_spp4_xorwow", "oracle_64_spp4_philox",
                                  "oracle_256_spp4_xorwow_rows", "oracle_64_spp16_xorwow_glm_b8"])
def test_matches_committed_fixture(pt, gpu, name):
    import os

    from conftest import GOLDEN

    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    size, spp, rng = int(g["size"]), int(g["spp"]), int(g["rng"])
    mb = 8 if name.endswith("_b8") else 5
    img, _ = pt.render_frame(size, size, spp, basis=g["basis"], eye=g["eye"], rng_mode=rng, max_bounces=mb)
    if "image" in g:
        assert_bit_exact(img, g["image"], name)
    else:
        assert_bit_exact(img[g["rows"]], g["row_data"], name)


# ---- every kernel variant computes the same bits ---------------------------------------------------
def _all_variants(pt, lab):
    """(module, variant) for every kernel variant: the product library's own (0, 6, 8, 9, 10, 13) from libptcore.so,
    the experimental and superseded ones (1-5, 7, 11, 12) from libptcore_lab.so."""
    prod = pt.variants()
    assert prod == [0, 6, 8, 9, 10, 13, 14] and lab.variants() == list(range(15))
    return [(pt, v) for v in prod] + [(lab, v) for v in lab.variants() if v not in prod]


@pytest.mark.parametrize("rng", [0, 1], ids=["xorwow", "philox"])
@pytest.mark.parametrize("spp", [1, 5, 8])
def test_all_variants_bit_exact_vs_oracle(pt, lab, oracle, gpu, rng, spp):
    """Every kernel variant, even / odd / single sample counts (variant 7 pairs samples), on the
    closed box and on an open scene where paths escape at different depths (variant 7 then has to
    retrace the second sample of a pair from the first one's true generator state)."""
    basis = pt.camera_basis(width=96, height=96)
    for name, sph in (("cornell", pt.scene_cornell()), ("open", pt.scene_cornell()[[0, 2, 4, 6, 7, 8]])):
        ref = oracle.render(96, 96, spp, spheres=sph, basis=basis, rng_mode=rng)
        for mod, v in _all_variants(pt, lab):
            img, _ = mod.render_frame(96, 96, spp, spheres=sph, basis=basis, rng_mode=rng, variant=v)
            assert_bit_exact(img, ref, f"{name} variant {v} rng {rng} spp {spp}")


# ---- multi-GPU tiling: a tile equals the same rows of the full frame --------------------------------
@pytest.mark.parametrize("rng", [0, 1], ids=["xorwow", "philox"])
def test_row_tiles_equal_full_frame(pt, oracle, gpu, rng):
    size, spp = 80, 4
    basis = pt.camera_basis(width=size, height=size)
    ref = oracle.render(size, size, spp, spheres=pt.scene_cornell(), basis=basis, rng_mode=rng)
    for b, e in ((0, 10), (10, 47), (47, 80)):
        tile, _ = pt.render_frame(size, size, spp, basis=basis, rng_mode=rng, row_begin=b, row_end=e)
        assert_bit_exact(tile, ref[b:e], f"rows [{b},{e}) rng {rng}")


# ---- generator state across frames (Renderer::d_states, pathtrace.cu:212,256) -----------------------
def test_xorwow_state_persists_across_frames_like_the_reference(pt, oracle, gpu):
    size, spp = 48, 2
    basis = pt.camera_basis(width=size, height=size)
    r = pt.Renderer(size, size, spp)
    d_scene, n = pt.upload_scene(pt.scene_cornell())
    d_out = pt.DeviceBuffer(size * size * 14 * 4)
    st = oracle.setup_random(size, size)
    assert np.array_equal(r.get_rng_state(), st)  # setup_random, pathtrace.cu:259-266
    for frame in range(3):
        r.render(d_out.ptr, d_scene.ptr, n, basis)
        img = d_out.download(np.float32, (size, size, 14))
        ref = oracle.render(size, size, spp, spheres=pt.scene_cornell(), basis=basis, rng_state=st)
        assert_bit_exact(img, ref, f"frame {frame}")
        assert np.array_equal(r.get_rng_state(), st)
    r.reset_rng()
    r.render(d_out.ptr, d_scene.ptr, n, basis)
    first = oracle.render(size, size, spp, spheres=pt.scene_cornell(), basis=basis)
    assert_bit_exact(d_out.download(np.float32, (size, size, 14)), first, "after reset_rng")
    r.destroy()


def test_philox_frames_are_keyed_not_stateful(pt, oracle, gpu):
    size, spp = 48, 2
    basis = pt.camera_basis(width=size, height=size)
    r = pt.Renderer(size, size, spp, rng_mode=pt.RNG_PHILOX, seed=0x1234ABCD5678)
    d_scene, n = pt.upload_scene(pt.scene_cornell())
    d_out = pt.DeviceBuffer(size * size * 14 * 4)
    for frame in (0, 1, 2):
        r.render(d_out.ptr, d_scene.ptr, n, basis)
        ref = oracle.render(size, size, spp, spheres=pt.scene_cornell(), basis=basis, rng_mode=1,
                            seed=0x1234ABCD5678, frame=frame)
        assert_bit_exact(d_out.download(np.float32, (size, size, 14)), ref, f"philox frame {frame}")
    r.set_frame(7)
    r.render(d_out.ptr, d_scene.ptr, n, basis)
    ref = oracle.render(size, size, spp, spheres=pt.scene_cornell(), basis=basis, rng_mode=1, seed=0x1234ABCD5678, frame=7)
    assert_bit_exact(d_out.download(np.float32, (size, size, 14)), ref, "philox frame 7")
    r.destroy()


# ---- BASELINE.json configs 4 and 5 at oracle-sized inputs, edge cases ---------------------------------
@pytest.mark.parametrize("with_walls", [True, False], ids=["closed", "open"])
def test_thousand_sphere_scene(pt, lab, oracle, gpu, with_walls):
    """config 4 shape: 1000 random spheres (LDS staging / intersect-loop stress); the open
    variant makes most paths escape at different depths (divergent early exit)."""
    sph = pt.scene_random(1000, seed=3, with_walls=with_walls)
    size, spp = 48, 2
    basis = pt.camera_basis(width=size, height=size)
    for mod, v in _all_variants(pt, lab):
        img, _ = mod.render_frame(size, size, spp, spheres=sph, basis=basis, variant=v)
        ref = oracle.render(size, size, spp, spheres=sph, basis=basis)
        assert_bit_exact(img, ref, f"1000 spheres walls={with_walls} variant {v}")


@pytest.mark.parametrize("rng", [0, 1], ids=["xorwow", "philox"])
def test_sweep_of_the_pooled_walk(pt, lab, oracle, gpu, rng):
    """Variant 13 leaves its lock-step DDA rounds for the sweep (pt_grid.h, grid_trips_pooled (2b)) once few lanes of a wave
    still walk: a full frame of complete waves over a 600-sphere scene (rays that stop in every cell of their way), closed and
    open, both generators, product and lab builds (two register allocations of the same source: EXACTNESS.md A.12)."""
    size = 64  # 4096 pixels: complete waves only, the sweep's precondition
    basis = pt.camera_basis(width=size, height=size)
    for walls in (True, False):
        sph = pt.scene_random(600, seed=11, with_walls=walls)
        ref = oracle.render(size, size, 2, spheres=sph, basis=basis, rng_mode=rng)
        for mod in (pt, lab):
            img, _ = mod.render_frame(size, size, 2, spheres=sph, basis=basis, rng_mode=rng, variant=13)
            assert_bit_exact(img, ref, f"sweep, 600 spheres walls={walls} {'lab' if mod is lab else 'product'}")


def test_sweep_crossings_with_equal_parameters(pt, lab, gpu):
    """The frame on which the sweep's first version lost a sphere (tools/grid_check.py, random150_open, third camera): a primary
    ray whose fourth x-crossing and ninth z-crossing of the grid have the same parameter to the last bit.  Each of the two
    crossings put itself first, and the cell behind both was never read.  Variant 13 against variant 10 (brute force; bit-exact
    against the oracle elsewhere) on the whole 1024 x 1024 x 8 spp frame, product and lab builds."""
    size, spp = 1024, 8
    scene = pt.scene_random(150, seed=150, with_walls=False)
    eye = (-150.0, 200.0, 500.0)
    basis = pt.camera_basis(eye, -55.0, -20.0, size, size)
    ref, _ = pt.render_frame(size, size, spp, spheres=scene, basis=basis, eye=eye, variant=10)
    assert np.count_nonzero(ref[..., 9]) > 5000, "the spheres are not in view"
    for mod in (pt, lab):
        img, _ = mod.render_frame(size, size, spp, spheres=scene, basis=basis, eye=eye, variant=13)
        assert_bit_exact(img, ref, f"variant 13 vs 10, random150_open camera 2 ({'lab' if mod is lab else 'product'})")


@pytest.mark.parametrize("rng", [0, 1], ids=["xorwow", "philox"])
def test_grid_walk_on_rays_whose_crossings_tie(pt, lab, oracle, gpu, rng):
    """Boundary crossings with EQUAL parameters, by construction: a scene that is symmetric under x <-> y and under the
    reflections about the eye's x and y (so is its grid: same origin, cell size and cell counts on both axes), a hand-made
    eye-ray basis with corners (+-0.3125, +-0.3125, -2) and a power-of-two image, 1 sample per pixel (no jitter): every operation that
    forms a primary ray's direction is exact, |d.x| = |d.y| to the last bit on both image diagonals, and on the main diagonal the
    x- and y-crossings of the grid tie at EVERY step (on the other one within roundings of the cell boundaries).  The lock-step
    DDA and the sweep must both walk a lattice path through such rays (EXACTNESS.md A.9 (viii): the first sweep lost the cell
    behind two tied crossings).  Then the same scene with jitter (2 spp): near-diagonal rays, near-ties."""
    g = np.random.default_rng(77)
    e = 50.0
    base = pt.scene_random(80, seed=31, with_walls=False)
    base["pos"][:, 0] = g.uniform(e + 1.0, e + 38.0, len(base)).astype(np.float32)
    base["pos"][:, 1] = g.uniform(e + 1.0, e + 38.0, len(base)).astype(np.float32)
    base["pos"][:, 2] = g.uniform(5.0, 95.0, len(base)).astype(np.float32)
    base["radius"][:] = g.uniform(1.5, 4.0, len(base)).astype(np.float32)
    parts = []
    for swap in (False, True):
        for fx in (False, True):
            for fy in (False, True):
                c = base.copy()
                x, y = (c["pos"][:, 1].copy(), c["pos"][:, 0].copy()) if swap else (c["pos"][:, 0].copy(), c["pos"][:, 1].copy())
                c["pos"][:, 0] = np.float32(2 * e) - x if fx else x
                c["pos"][:, 1] = np.float32(2 * e) - y if fy else y
                parts.append(c)
    scene = np.concatenate(parts)
    assert len(scene) == 640
    hdr = lab.grid_header(scene)
    assert hdr["valid"] == 1 and hdr["dims"][0] == hdr["dims"][1] and hdr["origin"][0] == hdr["origin"][1], hdr
    size = 128
    eye = (e, e, 300.0)
    k = 0.3125  # (dyadic: the corners, the pixel fractions and their products are exact)
    basis = np.array([-k, k, -2, k, k, -2, -k, -k, -2, k, -k, -2], dtype=np.float32)
    for spp in (1, 2):
        ref = oracle.render(size, size, spp, spheres=scene, basis=basis, eye=eye, rng_mode=rng)
        assert np.count_nonzero(ref[..., 9]) > size * size // 4, "the spheres are not in view"
        if spp == 1:  # the picture has the scene's symmetry on the main diagonal only if the rays there are exactly symmetric
            d = ref[..., 9]
            assert np.array_equal(d, d.T) or np.array_equal(d, d[::-1, ::-1].T), "the construction does not give symmetric rays"
        for mod in (pt, lab):
            img, _ = mod.render_frame(size, size, spp, spheres=scene, basis=basis, eye=eye, rng_mode=rng, variant=13)
            assert_bit_exact(img, ref, f"tied crossings, {spp} spp, {'lab' if mod is lab else 'product'}")


def test_interactive_shape_eight_bounces(pt, oracle, gpu):
    """config 5 shape: 4 spp per frame, 8-bounce cap, several frames into one device buffer."""
    size = 64
    basis = pt.camera_basis(width=size, height=size)
    r = pt.Renderer(size, size, 4, max_bounces=8)
    d_scene, n = pt.upload_scene(pt.scene_cornell())
    d_out = pt.DeviceBuffer(size * size * 14 * 4)
    st = oracle.setup_random(size, size)
    for frame in range(2):
        r.render(d_out.ptr, d_scene.ptr, n, basis)
        ref = oracle.render(size, size, 4, spheres=pt.scene_cornell(), basis=basis, max_bounces=8, rng_state=st)
        assert_bit_exact(d_out.download(np.float32, (size, size, 14)), ref, f"8 bounces frame {frame}")
    r.destroy()


def test_edge_cases(pt, lab, oracle, gpu):
    basis = pt.camera_basis(width=40, height=24)
    # non-square, ragged last workgroup (40*24 = 960 pixels = 3.75 workgroups)
    img, _ = pt.render_frame(40, 24, 3, basis=basis)
    assert_bit_exact(img, oracle.render(40, 24, 3, spheres=pt.scene_cornell(), basis=basis), "40x24")
    # empty scene: every ray escapes
    empty = pt.scene_cornell()[:0]
    img, _ = pt.render_frame(16, 16, 2, spheres=empty, basis=pt.camera_basis(width=16, height=16))
    assert np.all(img == 0)
    # max_bounces 0, a single pixel, a single sphere
    img, _ = pt.render_frame(16, 16, 2, basis=pt.camera_basis(width=16, height=16), max_bounces=0)
    assert np.all(img == 0)
    b1 = pt.camera_basis(width=1, height=1)
    img, _ = pt.render_frame(1, 1, 5, basis=b1)
    assert_bit_exact(img, oracle.render(1, 1, 5, spheres=pt.scene_cornell(), basis=b1), "1x1")
    one = pt.scene_cornell()[6:7]
    b32 = pt.camera_basis(width=32, height=32)
    img, _ = pt.render_frame(32, 32, 4, spheres=one, basis=b32)
    assert_bit_exact(img, oracle.render(32, 32, 4, spheres=one, basis=b32), "one sphere")
    # more samples than the table of sample-count reciprocals holds (1024): counts beyond it take the division, and an open
    # scene makes the colour accumulator's count lag the sample index (escaped paths do not update it, pathtrace.cu:157-161)
    b12 = pt.camera_basis(width=12, height=12)
    for name, sph in (("closed", pt.scene_cornell()), ("open", pt.scene_cornell()[[0, 2, 4, 6, 7, 8]])):
        for rng in (0, 1):
            for v in (None, 6, 8):
                img, _ = pt.render_frame(12, 12, 1300, spheres=sph, basis=b12, rng_mode=rng, variant=v)
                if v is None:
                    ref = oracle.render(12, 12, 1300, spheres=sph, basis=b12, rng_mode=rng)
                assert_bit_exact(img, ref, f"1300 spp {name} rng={rng} variant={v}")
    # sample chunking (frames of 512+ samples per pixel that make few rounds of workgroups: a pixel's samples go through
    # several workgroups of one launch, its state through HBM): ragged last workgroup, a row tile, two frames with the
    # generator state persisted, both generators, a sample count the chunks do not divide
    b40 = pt.camera_basis(width=40, height=24)
    for rng in (0, 1):
        r = pt.Renderer(40, 24, 601, rng_mode=rng, row_begin=3, row_end=20, variant=6)  # (the automatic choice for so small a tile is 8)
        d_scene, n = pt.upload_scene(pt.scene_cornell())
        d_out = pt.DeviceBuffer(17 * 40 * 56)
        st = oracle.setup_random(40, 24, row_begin=3, row_end=20) if rng == 0 else None
        for frame in range(2):
            r.render(d_out.ptr, d_scene.ptr, n, b40)
            ref = oracle.render(40, 24, 601, spheres=pt.scene_cornell(), basis=b40, rng_mode=rng, rng_state=st, frame=frame,
                                row_begin=3, row_end=20)
            assert_bit_exact(d_out.download(np.float32, (17, 40, 14)), ref, f"chunked 601 spp rng={rng} frame {frame}")
        assert r.kernel_info(n)["variant"] == 6
        r.destroy()
    # a zero-row tile is a no-op
    r = pt.Renderer(32, 32, 1, row_begin=5, row_end=5)
    d_scene, n = pt.upload_scene(pt.scene_cornell())
    assert r.render(None, d_scene.ptr, n, b32) == 0.0
    r.destroy()
    # too many spheres for the LDS staging budget: loud error, not a silent fallback
    # (the automatic choice has no limit: many-sphere scenes are not staged at all, see the next test)
    big = pt.scene_random(2500, seed=1)
    with pytest.raises(lab.PtError) as e:  # only the experimental variants still stage large scenes into LDS
        lab.render_frame(8, 8, 1, spheres=big, basis=pt.camera_basis(width=8, height=8), variant=5)
    assert e.value.code == -5
    with pytest.raises(pt.PtError) as e:  # and the product library says so when asked for one of them
        pt.Renderer(8, 8, 1, variant=5)
    assert e.value.code == -1 and "libptcore_lab" in str(e.value)


def test_ray_origins_on_sphere_surfaces(pt, lab, oracle, gpu):
    """The screen's doubt test (pt_intersect.h, screen_sphere_oc): an estimate is not trusted when the ray starts within
    rounding distance of a sphere's surface -- c = |o - centre|^2 - r^2 is zero or a few ulps of |o - centre|^2 -- where the
    small root's sign, i.e. WHICH root the reference returns, hangs on the rounding error of b*b.  Eyes placed exactly on,
    one ulp inside and one ulp outside sphere surfaces (unit, radius-600 and radius-1e5 spheres), looking inwards, outwards
    and along the tangent; every kernel family against the oracle."""
    size = 24
    scenes = []
    base = pt.scene_cornell()
    for radius, centre in ((1.0, (10.0, 20.0, 30.0)), (600.0, (50.0, 40.0, 80.0)), (1e5, (1e5 + 1.0, 40.8, 81.6))):
        sph = base.copy()
        sph["radius"][7] = radius
        sph["pos"][7] = centre
        scenes.append((radius, np.array(centre, dtype=np.float32), sph))
    many = pt.scene_random(300, seed=9, with_walls=True)  # the same through the grid and the many-sphere loop
    for radius, centre, sph in scenes:
        for axis in range(3):
            for nudge in (0, -1, 1, 4):
                eye = centre.copy()
                eye[axis] = np.float32(centre[axis] - np.float32(radius))
                bits = eye[axis:axis + 1].view(np.int32)
                bits += nudge if eye[axis] >= 0 else -nudge  # `nudge` ulps further from / closer to the centre
                for yaw, pitch in ((-90.0, 0.0), (90.0, 0.0), (0.0, 0.0), (-45.0, 30.0)):
                    e = tuple(float(x) for x in eye)
                    basis = pt.camera_basis(e, yaw, pitch, size, size)
                    ref = oracle.render(size, size, 2, spheres=sph, basis=basis, eye=e)
                    for v in (None, 8, 10):
                        img, _ = pt.render_frame(size, size, 2, spheres=sph, basis=basis, eye=e, variant=v)
                        assert_bit_exact(img, ref, f"eye on r={radius} axis {axis} nudge {nudge} yaw {yaw} variant {v}")
    radius, centre = float(many["radius"][10]), many["pos"][10].astype(np.float32)
    for nudge in (0, 1):
        eye = centre.copy()
        eye[0] = np.float32(centre[0] - np.float32(radius))
        eye[0:1].view(np.int32)[0] += nudge
        e = tuple(float(x) for x in eye)
        for yaw in (-90.0, 90.0, 0.0):
            basis = pt.camera_basis(e, yaw, 0.0, size, size)
            ref = oracle.render(size, size, 2, spheres=many, basis=basis, eye=e)
            for v in (None, 10, 11, 13):
                img, _ = (lab if v == 11 else pt).render_frame(size, size, 2, spheres=many, basis=basis, eye=e, variant=v)
                assert_bit_exact(img, ref, f"eye on a sphere of the 300-sphere scene, nudge {nudge} yaw {yaw} variant {v}")


@pytest.mark.parametrize("rng", [0, 1])
def test_planar_layout_is_the_transposed_frame(pt, lab, oracle, gpu, rng):
    """PT_LAYOUT_PLANAR writes [14][rows][width] (channel-first, coalesced without the LDS transpose); the values
    are the reference's, so the planes must equal the oracle's interleaved frame transposed -- every kernel
    family (one lane per pixel, four lanes per pixel, regeneration, grid), full frame and a row tile."""
    size, spp = 72, 5
    basis = pt.camera_basis(width=size, height=size)
    for scene, variants in ((pt.scene_cornell(), (0, 6, 8, 9, None)), (pt.scene_random(300, seed=4), (6, 8, 10, 11, 13, None))):
        ref = oracle.render(size, size, spp, spheres=scene, basis=basis, rng_mode=rng)
        for v in variants:
            mod = lab if v == 11 else pt
            img, _ = mod.render_frame(size, size, spp, spheres=scene, basis=basis, rng_mode=rng, variant=v, layout=pt.LAYOUT_PLANAR)
            planes = img.reshape(14, size, size)
            assert_bit_exact(np.ascontiguousarray(planes.transpose(1, 2, 0)), ref, f"planar variant={v}")
    tile, _ = pt.render_frame(size, size, spp, basis=basis, rng_mode=rng, row_begin=17, row_end=50, layout=pt.LAYOUT_PLANAR)
    full = oracle.render(size, size, spp, spheres=pt.scene_cornell(), basis=basis, rng_mode=rng)
    assert_bit_exact(np.ascontiguousarray(tile.reshape(14, 33, size).transpose(1, 2, 0)), full[17:50], "planar tile")
    with pytest.raises(pt.PtError):
        pt.Renderer(8, 8, 1, layout=7)


@pytest.mark.parametrize("rng", [0, 1])
def test_uniform_grid_variant(pt, lab, oracle, gpu, rng):
    """Variant 11 changes which spheres a lane tests (conservative uniform grid, rebuilt on the device every
    frame), never the result: against the oracle on closed/open scenes, a far-away camera (rays not admitted
    to the grid), radii spanning two decades (grid refused: too many spheres that cannot be registered) and a
    scene changed in place between two frames of one renderer."""
    size = 40
    cams = [((50.0, 52.0, 295.6), -90.0, 0.0), ((50.0, 40.0, 85.0), -60.0, 10.0), ((-900.0, 700.0, 2500.0), -70.0, -15.0)]
    scenes = {"walls": pt.scene_random(400, seed=21, with_walls=True), "open": pt.scene_random(400, seed=22, with_walls=False)}
    wide = pt.scene_random(400, seed=23, with_walls=True)
    wide["radius"][7:] = np.exp(np.random.default_rng(5).uniform(np.log(0.02), np.log(8.0), size=len(wide) - 7)).astype(np.float32)
    scenes["radii 0.02-8"] = wide
    for name, scene in scenes.items():
        for eye, yaw, pitch in cams:
            basis = pt.camera_basis(eye, yaw, pitch, size, size)
            ref = oracle.render(size, size, 3, spheres=scene, basis=basis, eye=eye, rng_mode=rng)
            for v in (13, None):  # the walk with the sphere tests pooled across the wave's lanes
                img, _ = pt.render_frame(size, size, 3, spheres=scene, basis=basis, eye=eye, rng_mode=rng, variant=v)
                assert_bit_exact(img, ref, f"grid {name} eye={eye} variant={v}")
            # the lab library: variant 11 (its predecessor: every lane tests its own spheres), variant 12 (the same walk decoupled
            # from the shading per lane; a measured negative result) and its own build of variant 13
            for v in (11, 12, 13):
                img, _ = lab.render_frame(size, size, 3, spheres=scene, basis=basis, eye=eye, rng_mode=rng, variant=v)
                assert_bit_exact(img, ref, f"grid {name} eye={eye} variant={v} (lab)")
    assert lab.grid_header(scenes["walls"])["valid"] == 1 and lab.grid_header(wide)["valid"] == 0
    r = pt.Renderer(size, size, 3, rng_mode=rng)
    assert r.kernel_info(400)["variant"] == 13
    # the grid belongs to the frame, not to the renderer: move the spheres between two frames
    basis = pt.camera_basis(width=size, height=size)
    a, b = scenes["walls"], scenes["walls"].copy()
    b["pos"][7:, 0] = 100.0 - b["pos"][7:, 0]
    d_scene, n = pt.upload_scene(a)
    d_out = pt.DeviceBuffer(size * size * 56)
    r.render(d_out.ptr, d_scene.ptr, n, basis)
    st = r.get_rng_state() if rng == 0 else None
    d_scene.upload(b)
    r.render(d_out.ptr, d_scene.ptr, n, basis)
    got = d_out.download(np.float32, (size, size, 14))
    r.destroy()
    ref = oracle.render(size, size, 3, spheres=b, basis=basis, rng_mode=rng, rng_state=st, frame=1)
    assert_bit_exact(got, ref, "scene changed in place between frames")


@pytest.mark.parametrize("rng", [0, 1])
def test_many_sphere_lean_lds_layout(pt, oracle, gpu, rng):
    """Scenes above PT_SCREEN_MAX_SPHERES are not staged into LDS: the sphere loop reads the caller's array
    with scalar loads, the winner's geometry and material are gathered per lane.  3000 spheres exceed what
    the LDS image could hold at all.  Automatic choice (variant 10) and the other two lean builds against
    the oracle, closed and open."""
    size = 48
    basis = pt.camera_basis(width=size, height=size)
    for walls in (True, False):
        scene = pt.scene_random(3000, seed=11, with_walls=walls)
        ref = oracle.render(size, size, 3, spheres=scene, basis=basis, rng_mode=rng)
        for v in (None, 6, 8, 10):
            img, _ = pt.render_frame(size, size, 3, spheres=scene, basis=basis, rng_mode=rng, variant=v)
            assert_bit_exact(img, ref, f"3000 spheres walls={walls} variant={v}")
    r = pt.Renderer(size, size, 3, rng_mode=rng)
    info = r.kernel_info(3000)
    r.destroy()
    assert info["variant"] == 10 and info["lds_bytes"] == 4 * 64 * 14 * 4 + 4880 and info["max_spheres"] == 1 << 26


# ---- full-size properties at BASELINE.json config 2 (1024 x 1024 x 1024 spp) ----------------------------
def test_full_size_config2_properties(pt, oracle, gpu):
    """At the headline size the oracle cannot render the whole frame in seconds, so: (1) three
    full rows are compared with the oracle bit for bit (3 x 1024 px x 1024 spp = 3.1 Msamples),
    rendered as 1-row tiles AND cut out of the full frame (tile independence at full size);
    (2) size-independent properties of the whole frame: finite, closed box => every pixel has a
    first hit so albedo/normal averages are bounded, variances non-negative."""
    size, spp = 1024, 1024
    basis = pt.camera_basis(width=size, height=size)
    r = pt.Renderer(size, size, spp, persist_rng=False)
    d_scene, n = pt.upload_scene(pt.scene_cornell())
    d_out = pt.DeviceBuffer(size * size * 14 * 4)
    ms = r.render(d_out.ptr, d_scene.ptr, n, basis)
    full = d_out.download(np.float32, (size, size, 14))
    r.destroy()
    assert np.isfinite(full).all()
    assert (full[..., 10:] >= 0).all()
    nrm = np.linalg.norm(full[..., 3:6], axis=-1)
    assert (nrm <= 1.0 + 1e-3).all() and (full[..., 9] > 0).all()  # 1024 float32 adds per component
    assert (full[..., 6:9] >= 0).all() and (full[..., 6:9] <= 1.0).all()
    for row in (0, 511, 1023):
        ref = oracle.render(size, size, spp, spheres=pt.scene_cornell(), basis=basis, row_begin=row, row_end=row + 1)
        assert_bit_exact(full[row:row + 1], ref, f"full frame row {row}")
        tile, _ = pt.render_frame(size, size, spp, basis=basis, row_begin=row, row_end=row + 1, persist_rng=False)
        assert_bit_exact(tile, ref, f"tile row {row}")
    print(f"config 2 frame: {ms:.1f} ms, {size * size * spp / ms / 1e3:.0f} Msamples/s")


# ---- the C++ look-alike classes + CLI, end to end -------------------------------------------------------
def _read_feature_exr(path):
    """Minimal reader for the 14-channel uncompressed scanline EXR the writer emits."""
    import struct

    raw = open(path, "rb").read()
    assert raw[:4] == bytes([0x76, 0x2F, 0x31, 0x01])
    pos, names, w, h = 8, [], None, None
    while raw[pos] != 0:
        e = raw.index(b"\0", pos); name = raw[pos:e].decode(); pos = e + 1
        e = raw.index(b"\0", pos); typ = raw[pos:e].decode(); pos = e + 1
        (ln,) = struct.unpack("<I", raw[pos:pos + 4]); pos += 4
        val = raw[pos:pos + ln]; pos += ln
        if name == "channels":
            q = 0
            while val[q] != 0:
                e = val.index(b"\0", q); names.append(val[q:e].decode()); q = e + 1 + 16
        if name == "dataWindow":
            x0, y0, x1, y1 = struct.unpack("<4i", val); w, h = x1 - x0 + 1, y1 - y0 + 1
        assert typ
    pos += 1
    offs = struct.unpack(f"<{h}Q", raw[pos:pos + 8 * h])
    planes = np.zeros((len(names), h, w), dtype=np.float32)
    for y in range(h):
        o = offs[y] + 8
        planes[:, y, :] = np.frombuffer(raw[o:o + 4 * w * len(names)], dtype="<f4").reshape(len(names), w)
    return names, planes


def test_cli_front_end_renders_and_saves_like_main_cu(pt, oracle, gpu, tmp_path):
    import os
    import subprocess

    from conftest import ROOT

    exe = os.path.join(ROOT, "cuda-pathtrace_amd", "pathtrace")
    assert os.path.exists(exe), "build it with __graft_entry__.build()"
    out = str(tmp_path / "frame")
    res = subprocess.run([exe, "--size", "64", "-s", "4", "-o", out], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr
    assert "cuda-pathtrace 0.3" in res.stdout and "Render completed in" in res.stdout and "fps)" in res.stdout
    names, planes = _read_feature_exr(out + ".exr")
    basis = pt.camera_basis(width=64, height=64)
    ref = oracle.render(64, 64, 4, spheres=pt.scene_cornell(), basis=basis)
    src = {"Color.R": 0, "Color.G": 1, "Color.B": 2, "Normal.X": 3, "Normal.Y": 4, "Normal.Z": 5, "Albedo.R": 6,
           "Albedo.G": 7, "Albedo.B": 8, "Depth.Z": 9, "ColorVar.Z": 10, "NormalVar.Z": 11, "AlbedoVar.Z": 12,
           "DepthVar.Z": 13}
    assert len(names) == 14
    for k, nm in enumerate(names):
        assert np.array_equal(planes[k].view(np.uint32), ref[..., src[nm]].view(np.uint32)), nm
    for suffix in ("_color", "_normal", "_albedo", "_depth", "_color_var", "_normal_var", "_albedo_var", "_depth_var"):
        assert os.path.getsize(out + suffix + ".bmp") == 54 + 64 * 64 * 3
    # error path: gpuErrchk look-alike prints GPUassert and exits non-zero
    bad = subprocess.run([exe, "--size", "16", "-s", "1", "--device", "99", "-o", out], capture_output=True, text=True)
    assert bad.returncode != 0 and "GPUassert:" in bad.stderr


def test_cli_fly_through_matches_frame_by_frame_oracle(pt, oracle, gpu, tmp_path):
    """--poses: the headless interactive loop (main.cu:146-148): camera moves every frame, renderer
    and generator state persist.  The saved last frame must equal the oracle driven the same way."""
    import os
    import subprocess

    from conftest import ROOT

    poses = [(50.0, 52.0, 295.6, -90.0, 0.0), (48.0, 50.0, 280.0, -88.0, -2.0), (55.0, 45.0, 250.0, -95.0, 3.0)]
    pf = tmp_path / "poses.txt"
    pf.write_text("".join("%g %g %g %g %g\n" % p for p in poses))
    out = str(tmp_path / "fly")
    exe = os.path.join(ROOT, "cuda-pathtrace_amd", "pathtrace")
    res = subprocess.run([exe, "--size", "64", "-s", "4", "--max-bounces", "8", "--poses", str(pf), "--nobitmap", "-o", out],
                         capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr
    assert "Fly-through: 3 frames" in res.stdout
    st = oracle.setup_random(64, 64)
    for x, y, z, yaw, pitch in poses:
        basis = pt.camera_basis((x, y, z), yaw, pitch, 64, 64)
        ref = oracle.render(64, 64, 4, spheres=pt.scene_cornell(), basis=basis, eye=(x, y, z), max_bounces=8, rng_state=st)
    names, planes = _read_feature_exr(out + ".exr")
    k = names.index("Color.R")
    assert np.array_equal(planes[k].view(np.uint32), ref[..., 0].view(np.uint32))
    k = names.index("DepthVar.Z")
    assert np.array_equal(planes[k].view(np.uint32), ref[..., 13].view(np.uint32))


# ---- fuzz: random scenes and cameras, every variant ----------------------------------------------------
def _random_scene(rng, n):
    s = np.zeros(n, dtype=np.dtype([("radius", "<f4"), ("pos", "<f4", 3), ("emission", "<f4", 3), ("color", "<f4", 3)]))
    kind = rng.integers(0, 4, n)
    for i in range(n):
        if kind[i] == 0:      # small sphere somewhere in front of the camera
            s["radius"][i] = rng.uniform(0.5, 20.0)
            s["pos"][i] = rng.uniform([0, 0, 0], [100, 80, 170])
        elif kind[i] == 1:    # huge "wall" sphere the camera is inside of
            r = 10.0 ** rng.uniform(3, 5)
            axis = rng.integers(0, 3)
            sign = rng.choice([-1.0, 1.0])
            p = np.array([50.0, 40.0, 80.0])
            p[axis] += sign * (r - rng.uniform(20, 120))
            s["radius"][i], s["pos"][i] = r, p
        elif kind[i] == 2:    # sphere containing the camera position region partially (origin near surface)
            s["radius"][i] = rng.uniform(30, 300)
            s["pos"][i] = rng.uniform([-100, -100, 0], [200, 200, 400])
        else:                 # concentric / coincident centres: exact ties between spheres
            j = rng.integers(0, max(i, 1))
            s["pos"][i] = s["pos"][j]
            s["radius"][i] = s["radius"][j] if rng.random() < 0.5 else s["radius"][j] * rng.uniform(0.5, 1.5)
        s["color"][i] = rng.uniform(0, 1, 3)
        s["emission"][i] = rng.uniform(0, 5, 3) if rng.random() < 0.2 else 0.0
    return s


@pytest.mark.parametrize("seed", range(48))
def test_fuzz_random_scenes_all_variants(pt, lab, oracle, gpu, seed):
    """Random sphere soups (huge walls, nested and DUPLICATED spheres = exact ties, origins near
    surfaces), random cameras, both generators: every variant must equal the oracle bit for bit.
    Duplicated spheres force the first-index tie-break and the 'ambiguous -> literal loop' path."""
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(1, 40)) if seed % 4 else int(rng.integers(65, 120))  # also scenes above the key-based screen's 64-sphere limit
    scene = _random_scene(rng, n)
    size = 48
    eye = tuple(rng.uniform([20, 20, 100], [80, 60, 300]))
    basis = pt.camera_basis(eye, float(rng.uniform(-120, -60)), float(rng.uniform(-20, 20)), size, size)
    mode = int(seed % 2)
    spp = int(rng.integers(1, 7))
    mb = int(rng.integers(1, 9))
    ref = oracle.render(size, size, spp, spheres=scene, basis=basis, eye=eye, rng_mode=mode, max_bounces=mb)
    assert np.isfinite(ref).all()
    for mod, v in _all_variants(pt, lab):
        img, _ = mod.render_frame(size, size, spp, spheres=scene, basis=basis, eye=eye, rng_mode=mode, max_bounces=mb, variant=v)
        assert_bit_exact(img, ref, f"fuzz seed {seed} n={n} spp={spp} bounces={mb} variant {v}")


def test_automatic_variant_policy(pt, lab, oracle, gpu):
    """Default options: small tiles use the four-lanes-per-pixel kernel (variant 8); a renderer whose
    scene turns out to be open (speculation keeps failing) goes back to variant 6; large tiles use
    variant 6 from the start.  Whatever is chosen, frames stay bit-identical to the oracle."""
    size, spp = 256, 8
    basis = pt.camera_basis(width=size, height=size)
    d_out = pt.DeviceBuffer(size * size * 14 * 4)
    for name, sph, expect_after in (("closed", pt.scene_cornell(), 8), ("open", pt.scene_cornell()[[6, 7, 8]], 6)):
        r = pt.Renderer(size, size, spp)  # variant left to the library
        d_scene, n = pt.upload_scene(sph)
        assert r.kernel_info(n)["variant"] == 8, name
        st = oracle.setup_random(size, size)
        for frame in range(3):
            r.render(d_out.ptr, d_scene.ptr, n, basis)
            ref = oracle.render(size, size, spp, spheres=sph, basis=basis, rng_state=st)
            assert_bit_exact(d_out.download(np.float32, (size, size, 14)), ref, f"{name} frame {frame}")
        assert r.kernel_info(n)["variant"] == expect_after, name
        r.destroy()
    big = pt.Renderer(1024, 1024, 8)
    assert big.kernel_info(9)["variant"] == 6
    # many-sphere scenes: the grid kernel from 160 spheres on any tile, from 72 on a tile that fills the chip (tools/grid_threshold.py)
    # ... with 1024-thread workgroups (variant 14) above 1200 spheres (tools/large_scenes.py)
    assert [big.kernel_info(n)["variant"] for n in (11, 71, 72, 159, 160, 1200, 1201, 2048, 2049)] == [10, 10, 13, 13, 13, 13, 14, 14, 10]
    big.destroy()
    small = pt.Renderer(128, 128, 8)
    assert [small.kernel_info(n)["variant"] for n in (71, 72, 159, 160)] == [8, 8, 8, 13]
    small.destroy()
    size = 512  # a frame of four one-lane waves per SIMD: the automatic choice for 100 spheres is the grid kernel
    sph = pt.scene_random(100, seed=41, with_walls=True)
    basis = pt.camera_basis(width=size, height=size)
    img, _ = pt.render_frame(size, size, 1, spheres=sph, basis=basis)
    assert_bit_exact(img, oracle.render(size, size, 1, spheres=sph, basis=basis), "100 spheres, automatic (grid) kernel")
    # Which of variants 6 / 8 / 9 a small scene gets is decided by the cost model (csrc/pt_capi.hip; tests/test_policy_model.py
    # holds it against the measured sweeps): the expectations below are the MODEL's, computed through the lab library, not
    # literals -- row tiles of the headline frame (what a rank of a multi-GPU run renders) and short frames, both generators.
    simds = gpu["compute_units"] * 4

    def expect(width, rows, spp, rng=0, bounces=5, with9=True):
        return lab.policy_choice(rng, width * rows / 64.0 / simds, spp, bounces, with9)

    for rng in (0, 1):
        for rows in (64, 128, 160, 192, 224, 256, 320, 512, 1024):
            r = pt.Renderer(1024, 1024, 1024, row_begin=0, row_end=rows, rng_mode=rng)
            assert r.kernel_info(9)["variant"] == expect(1024, rows, 1024, rng), (rng, rows)
            r.destroy()
        for size, spp, mb in ((320, 64, 5), (320, 16, 5), (576, 64, 5), (576, 16, 5), (640, 64, 5), (512, 4, 8), (256, 4, 8), (256, 4, 5), (1024, 8, 5)):
            r = pt.Renderer(size, size, spp, max_bounces=mb, rng_mode=rng)
            assert r.kernel_info(9)["variant"] == expect(size, size, spp, rng, mb), (rng, size, spp, mb)
            r.destroy()
    assert expect(1024, 1024, 1024) == 6 and expect(1024, 64, 1024) == 8 and expect(512, 512, 4, 0, 8) == 6  # (what the model says there)
    r = pt.Renderer(1024, 1024, 1024, row_begin=0, row_end=128, max_bounces=6)  # no reference-configuration build: variant 9 is out
    assert r.kernel_info(9)["variant"] == expect(1024, 128, 1024, 0, 6, with9=False) == 8
    r.destroy()
    few = pt.Renderer(256, 256, 2)  # too few samples to split
    assert few.kernel_info(9)["variant"] == 6
    few.destroy()


@pytest.mark.parametrize("variant", [8, 9], ids=["four_lanes", "two_lanes"])
@pytest.mark.parametrize("scene", ["closed", "open"])
@pytest.mark.parametrize("spp", [3, 8, 13])
def test_four_lane_kernel_keeps_generator_state_across_frames(pt, oracle, gpu, scene, spp, variant):
    """Variants 8 and 9 forced: the generator state written back after a frame must be the sequential one
    (also when speculation failed and the pixel finished in sequential mode, and when spp is not a
    multiple of the lanes per pixel), so later frames continue the reference's stream."""
    size = 40
    sph = pt.scene_cornell() if scene == "closed" else pt.scene_cornell()[[1, 3, 6, 7, 8]]
    basis = pt.camera_basis(width=size, height=size)
    r = pt.Renderer(size, size, spp, variant=variant)
    d_scene, n = pt.upload_scene(sph)
    d_out = pt.DeviceBuffer(size * size * 14 * 4)
    st = oracle.setup_random(size, size)
    for frame in range(3):
        r.render(d_out.ptr, d_scene.ptr, n, basis)
        ref = oracle.render(size, size, spp, spheres=sph, basis=basis, rng_state=st)
        assert_bit_exact(d_out.download(np.float32, (size, size, 14)), ref, f"{scene} spp {spp} frame {frame}")
        assert np.array_equal(r.get_rng_state(), st), f"{scene} spp {spp}: generator state after frame {frame}"
    r.destroy()


def test_display_packer_matches_denoise_kernel(pt, oracle, gpu):
    """pt_display_pack (the reference's Denoiser::Denoise / denoise_kernel, src/denoise.cu:9-29):
    bit-exact vs the oracle on a rendered frame and on values that exercise the clamp
    (negative, > 1, NaN, exact 1/255 steps)."""
    basis = pt.camera_basis(width=64, height=64)
    img, _ = pt.render_frame(64, 64, 4, basis=basis)
    assert np.array_equal(pt.display_pack(img).view(np.uint32), oracle.display_pack(img).view(np.uint32))
    rng = np.random.default_rng(3)
    syn = rng.uniform(-0.5, 1.5, (33, 47, 14)).astype(np.float32)
    syn[0, 0, :3] = [np.nan, -0.0, 1.0]
    syn[1, :, 0] = np.arange(47, dtype=np.float32) / 255.0
    syn[2, :, 1] = np.nextafter(np.arange(47, dtype=np.float32) / np.float32(255.0), np.float32(0))
    a, b = pt.display_pack(syn), oracle.display_pack(syn)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert a[5, 7, 0] == 7 and a[5, 7, 1] == 47 - 5          # (col, width - row)
    assert (a[..., 2].view(np.uint32) >> 24 == 1).all()      # alpha byte = 1


def test_fused_display_vertices_equal_denoise_of_the_frame(pt, oracle, gpu):
    """pt_renderer_set_display: Renderer::Render and Denoiser::Denoise (main.cu:148,175; src/denoise.cu:9-29) in ONE kernel.  The
    vertices every frame then writes must be what the oracle's denoise_kernel makes of the oracle's frame, bit for bit -- for
    the interactive shape (BASELINE configs[4]: 8 bounces, 4 spp, consecutive frames with the generator state carried over),
    for every kernel family (one, two and four lanes per pixel, chunked, path regeneration, the pooled grid walk, both
    generators), for a row tile with ragged waves, and switched off again by NULL; the fast mode's vertices are the display
    pack of the fast mode's own frame."""
    def run(w, h, spp, scene, frames=1, tile=None, **kw):
        basis = pt.camera_basis(width=w, height=h)
        rb, re_ = tile if tile else (0, h)
        r = pt.Renderer(w, h, spp, row_begin=rb if tile else 0, row_end=re_ if tile else 0, **kw)
        d_scene, n = pt.upload_scene(scene)
        d_out, d_vtx = pt.DeviceBuffer((re_ - rb) * w * 56), pt.DeviceBuffer((re_ - rb) * w * 12)
        r.set_display(d_vtx.ptr)
        rng_mode = kw.get("rng_mode", 0)
        st = oracle.setup_random(w, h) if rng_mode == 0 and not kw.get("fast_math") else None
        for f in range(frames):
            d_vtx.upload(np.full(((re_ - rb), w, 3), -3.0, dtype=np.float32))
            r.render(d_out.ptr, d_scene.ptr, n, basis)
            got = d_vtx.download(np.float32, (re_ - rb, w, 3))
            frame = d_out.download(np.float32, (re_ - rb, w, 14))
            if kw.get("fast_math"):
                want = pt.display_pack(frame)  # (no tile in this case)
            else:
                ref = oracle.render(w, h, spp, spheres=scene, basis=basis, rng_mode=rng_mode, rng_state=st, frame=f,
                                    max_bounces=kw.get("max_bounces", 5))
                assert_bit_exact(frame, ref[rb:re_], f"{kw} frame {f}")
                want = oracle.display_pack(ref)[rb:re_]
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), f"{kw} tile {tile} frame {f}"
        r.set_display(None)
        d_vtx.upload(np.full(((re_ - rb), w, 3), -3.0, dtype=np.float32))
        r.render(d_out.ptr, d_scene.ptr, n, basis)
        assert (d_vtx.download(np.float32, (re_ - rb, w, 3)) == -3.0).all()
        r.destroy()

    cornell = pt.scene_cornell()
    run(512, 512, 4, cornell, frames=2, max_bounces=8)                      # the interactive shape, automatic kernel (REF8 build)
    for rng in (0, 1):
        for v in (0, 6, 8, 9):
            run(72, 40, 9, cornell, variant=v, rng_mode=rng)
        run(72, 40, 601, cornell, variant=6, rng_mode=rng, chunks=4)         # the last chunk writes frame and vertices
        run(72, 40, 5, pt.scene_random(40, seed=2), variant=10, rng_mode=rng)
        run(72, 40, 5, pt.scene_random(300, seed=3), variant=13, rng_mode=rng)
    run(100, 60, 4, cornell, tile=(17, 43), variant=6)                         # a row tile: vertices carry the IMAGE row
    run(100, 60, 4, cornell, tile=(17, 43), variant=8)
    run(96, 64, 4, cornell, fast_math=True)


def test_cli_preview_is_the_display_packed_frame(pt, oracle, gpu, tmp_path):
    import os
    import subprocess

    from conftest import ROOT

    out, ppm = str(tmp_path / "f"), str(tmp_path / "f.ppm")
    exe = os.path.join(ROOT, "cuda-pathtrace_amd", "pathtrace")
    res = subprocess.run([exe, "--size", "32", "-s", "8", "--nobitmap", "-o", out, "--preview", ppm], capture_output=True,
                         text=True, timeout=120)
    assert res.returncode == 0, res.stderr
    raw = open(ppm, "rb").read()
    assert raw.startswith(b"P6\n32 32\n255\n")
    rgb = np.frombuffer(raw[len(b"P6\n32 32\n255\n"):], dtype=np.uint8).reshape(32, 32, 3)
    ref = oracle.render(32, 32, 8, spheres=pt.scene_cornell(), basis=pt.camera_basis(width=32, height=32))
    want = oracle.display_pack(ref)[..., 2].copy().view(np.uint8).reshape(32, 32, 4)[..., :3]
    assert np.array_equal(rgb, want)


def test_wide_grid_kernel_on_large_scenes(pt, oracle, gpu):
    """Variant 14 = variant 13 with 1024-thread workgroups (one per CU: the cell table of a scene above ~1200 spheres gets the other half
    of the LDS).  1500 and 2048 spheres, closed and open, both generators, a frame of ragged 1024-pixel blocks; and the automatic
    policy picks it for such scenes, variant 13 up to 1200 spheres."""
    w, h, spp = 200, 88, 4  # 17 600 pixels: 17 full workgroups and one with 192 of its 1024 lanes
    basis = pt.camera_basis(width=w, height=h)
    for n, walls, rng in ((1500, True, 0), (2048, False, 1), (2048, True, 0)):
        sph = pt.scene_random(n, seed=21, with_walls=walls)
        ref = oracle.render(w, h, spp, spheres=sph, basis=basis, rng_mode=rng)
        for v in (14, 13):
            img, _ = pt.render_frame(w, h, spp, spheres=sph, basis=basis, rng_mode=rng, variant=v)
            assert_bit_exact(img, ref, f"{n} spheres walls={walls} rng {rng} variant {v}")
    r = pt.Renderer(1024, 1024, 4)
    assert r.kernel_info(1500)["variant"] == 14 and r.kernel_info(1500)["block_threads"] == 1024
    assert r.kernel_info(1000)["variant"] == 13 and r.kernel_info(2049)["variant"] == 10
    r.destroy()


@pytest.mark.parametrize("rng", [0, 1], ids=["xorwow", "philox"])
def test_last_bounce_shortcut_of_the_grid_kernel(pt, oracle, gpu, rng):
    """Variants 13 / 14 skip the walk of a path's LAST bounce when the ranking of the spheres outside the grid plus the (<= 32) emitting
    spheres inside it is won, clear of the ambiguity margin, by a sphere that does not emit (csrc/pt_grid.h, PT_V13_LAST_SHORTCUT;
    EXACTNESS.md A.16).  Scenes built to sit on every branch of that rule: no / 12 / exactly 32 / 33 emitting grid spheres (33: rule
    off), an emitting WALL, NaN and negative emission, emitting spheres packed around the camera so that last bounces often end on
    them, no walls at all (nothing certifies a hit); 2..8 bounces; against the oracle, bit for bit."""
    g = np.random.default_rng(1234 + rng)
    w, h = 128, 64
    basis = pt.camera_basis(width=w, height=h)
    cases = []
    for n_em, walls, wall_emits, special in ((0, True, False, None), (12, True, False, None), (32, True, True, None), (33, True, False, None),
                                             (12, False, False, None), (10, True, False, "nan"), (14, True, False, "near")):
        sc = pt.scene_random(600, seed=100 + n_em, with_walls=walls)
        k0 = 7 if walls else 0
        sc["emission"][k0:] = 0.0
        pick = k0 + g.choice(len(sc) - k0, size=n_em, replace=False)
        sc["emission"][pick] = g.uniform(0.5, 6.0, size=(n_em, 3)).astype(np.float32)
        if wall_emits:
            sc["emission"][1] = (0.3, 0.2, 0.1)
        if special == "nan":
            sc["emission"][pick[0]] = (np.nan, 1.0, 0.0)
            sc["emission"][pick[1]] = (-2.0, 0.0, 0.0)
            sc["emission"][pick[2]] = (0.0, -0.0, np.inf)
        if special == "near":  # the emitting spheres in a shell around the default camera's view: many last bounces end on them
            sc["pos"][pick] = (np.array([50.0, 45.0, 120.0]) + g.normal(0, 18.0, size=(n_em, 3))).astype(np.float32)
            sc["radius"][pick] = g.uniform(2.5, 3.0, size=n_em).astype(np.float32)
        cases.append((f"{n_em} emitting, walls={walls}, wall emits={wall_emits}, {special}", sc))
    for name, sc in cases:
        for mb, spp in ((5, 4), (2, 5), (8, 4), (3, 6)):
            ref = oracle.render(w, h, spp, spheres=sc, basis=basis, rng_mode=rng, max_bounces=mb)
            if "nan" in name:
                # a NaN's sign and payload are the machine's (x86 and gfx950 produce different quiet NaNs from the same operation; the
                # contract fixes values): which floats are NaN must agree with the oracle, everything else bit for bit -- and the grid
                # kernels must agree with the brute-force kernel on the same machine in every bit
                brute, _ = pt.render_frame(w, h, spp, spheres=sc, basis=basis, rng_mode=rng, max_bounces=mb, variant=10)
                assert np.array_equal(np.isnan(brute), np.isnan(ref)) and np.isnan(ref).any()
                ok = ~np.isnan(ref)
                assert np.array_equal(brute.view(np.uint32)[ok], ref.view(np.uint32)[ok])
                ref = brute
            for v in (13, 14):
                img, _ = pt.render_frame(w, h, spp, spheres=sc, basis=basis, rng_mode=rng, max_bounces=mb, variant=v)
                assert_bit_exact(img, ref, f"{name}; {mb} bounces, {spp} spp, variant {v}")
