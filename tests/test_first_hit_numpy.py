"""First-hit channels of a 1-spp frame from a SECOND, independent restatement of the reference's lines -- vectorised numpy with
explicit float32 / float64 dtypes (tests/numpy_restatement.py), written from src/pathtrace.cu and sharing no code with
oracle/pt_oracle.c or the HIP kernels.

With spp == 1 the primary ray has no jitter (src/pathtrace.cu:222), so normal, albedo and depth of a pixel (:188-191) are a
function of the camera basis and the scene only -- no random number reaches them.  What they exercise is exactly the arithmetic the
contract is most particular about: the lerp of the ray basis (:229), intersectSphere with its float / double promotions
(:72-91: `2.0 * dot`, `b*b - 4 * a*c` in float against `(b*b) - 4.0*a*c` in double, the division by `2.0*a`, the rounding of tNear
and tFar to float), the strict `t > 0 && t < tNearest` with first-index-wins of intersectScene (:93-107), the hit point and the
normal (:163-166; normalize = v * rsqrtf(dot), rsqrtf := 1/sqrtf -- contract C2, helper_math's host path).  Every operation below
is one numpy ufunc on arrays of a stated dtype, i.e. one IEEE rounding -- contract C1 (no contraction) by construction.
Compared BIT FOR BIT (channels 3..9; 11..13 are 0 at spp 1, :61-62) with the oracle here and with every HIP kernel family (-m gpu)."""
import numpy as np
import pytest

from numpy_restatement import EYE, first_hit_frame  # noqa: E402  (tests/ is on sys.path: conftest)

f32 = np.float32


def _scenes(mod):
    rs = np.random.default_rng(11)
    cornell = mod.scene_cornell()
    n = 40
    few = np.zeros(n + 1, dtype=mod.SPHERE_DTYPE)  # small spheres in the view frustum (the default camera looks down -z)
    for k in range(n):
        few[k]["pos"] = np.asarray(EYE, dtype=np.float64) + (rs.normal() * 45.0, rs.normal() * 35.0, -rs.uniform(60.0, 420.0))
        few[k]["radius"] = rs.uniform(2.0, 16.0)
        few[k]["color"] = rs.uniform(0.1, 0.9, size=3)
    few[n]["pos"] = np.asarray(EYE, dtype=np.float64) + (5.0, -3.0, 20.0)  # LAST in the list: a sphere around the eye (tFar, the flip of :166)
    few[n]["radius"] = 700.0
    few[n]["color"] = (0.5, 0.25, 0.125)
    return {"cornell": cornell, "random_open": few[:n].copy(), "random_enclosed": few}


SCENES = ["cornell", "random_open", "random_enclosed"]


def test_the_scenes_cover_misses_many_spheres_and_hits_from_inside(oracle):
    sc = _scenes(oracle)
    basis = oracle.camera_basis(w=96, h=96)
    open_ = first_hit_frame(96, 96, sc["random_open"], basis)
    hit = open_[..., 6] > 0
    assert 0.2 < hit.mean() < 0.9                                                  # rays that leave the scene (:157-161)
    assert len(np.unique(open_[..., 3:6].reshape(-1, 3), axis=0)) > 12             # many different nearest spheres
    enclosed = first_hit_frame(96, 96, sc["random_enclosed"], basis)
    assert np.all(enclosed[..., 6] > 0)
    assert np.array_equal(enclosed[hit], open_[hit]) and np.all(enclosed[~hit][:, 3] == np.float32(0.5))  # the far side of the last sphere


def _check(img, want, what):
    got = np.ascontiguousarray(img[..., 3:10]).view(np.uint32)
    exp = np.ascontiguousarray(want).view(np.uint32)
    bad = np.argwhere(got != exp)
    assert bad.size == 0, f"{what}: {len(bad)} floats differ, first at (row, col, channel) {bad[0]}: " \
                          f"got {img[bad[0][0], bad[0][1], 3 + bad[0][2]]!r} want {want[tuple(bad[0])]!r}"
    assert np.all(img[..., 11:14] == 0.0)  # one sample: no variance (:61-62)


@pytest.mark.parametrize("size", [64, 96, 256])  # a power of two (x / w exact) and not
@pytest.mark.parametrize("scene", SCENES)
def test_oracle_first_hit_channels_equal_the_numpy_restatement(oracle, scene, size):
    spheres = _scenes(oracle)[scene]
    basis = oracle.camera_basis(w=size, h=size)
    want = first_hit_frame(size, size, spheres, basis)
    assert want[..., 6].max() > 0.0
    for mode in (oracle.RNG_XORWOW, oracle.RNG_PHILOX):
        img = oracle.render(size, size, 1, spheres, basis, max_bounces=2, rng_mode=mode, threads=4)
        _check(img, want, f"oracle {scene} {size}")


def test_the_comparison_is_sensitive_to_the_promotions(oracle):
    """Reading :80-81 in float instead of double changes depths of the very frames compared above: the equality is not vacuous."""
    basis = oracle.camera_basis(w=256, h=256)
    spheres = oracle.scene_cornell()
    img = oracle.render(256, 256, 1, spheres, basis, max_bounces=1, threads=4)
    wrong = first_hit_frame(256, 256, spheres, basis, promote=False)
    n_diff = int(np.count_nonzero(np.ascontiguousarray(img[..., 3:10]).view(np.uint32) != wrong.view(np.uint32)))
    assert n_diff > 1000, n_diff


def test_numpy_restatement_hits_the_survey_pixels(oracle):
    """SURVEY.md 8(c)'s four recorded 1-spp pixels (tests/golden/survey_kats.json), to the digits recorded there."""
    import json
    import os
    import sys

    from conftest import GOLDEN

    sys.path.insert(0, GOLDEN)
    from make_golden import closed_form_basis  # (the basis those values were recorded with)

    kats = json.load(open(os.path.join(GOLDEN, "survey_kats.json")))["aov_256_spp1"]
    want = first_hit_frame(256, 256, oracle.scene_cornell(), closed_form_basis(256, 256))
    for key, rec in kats.items():
        if not isinstance(rec, dict):
            continue
        r, c = (int(v) for v in key.split(","))
        np.testing.assert_allclose(want[r, c, 0:3], rec["normal"], rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose(want[r, c, 3:6], rec["albedo"], rtol=1e-6)
        np.testing.assert_allclose(want[r, c, 6], rec["depth"], rtol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("scene", SCENES)
def test_hip_first_hit_channels_equal_the_numpy_restatement(pt, gpu, scene):
    spheres = _scenes(pt)[scene]
    for size in (96, 256):
        basis = pt.camera_basis(width=size, height=size)
        want = first_hit_frame(size, size, spheres, basis)
        for rng in (pt.RNG_XORWOW, pt.RNG_PHILOX):
            for variant in (None, 0, 6, 8, 10):  # automatic, literal, screened, four lanes per pixel, regeneration
                img, _ = pt.render_frame(size, size, 1, spheres, basis, max_bounces=2, rng_mode=rng, variant=variant)
                _check(img, want, f"HIP {scene} {size} variant {variant}")


@pytest.mark.gpu
def test_hip_grid_kernels_first_hit_channels_equal_the_numpy_restatement(pt, lab, gpu):
    """The many-sphere kernels (uniform grid -- variant 11, lab library; pooled tests; 1024-thread workgroups) on BASELINE config 4's kind of scene."""
    size = 128
    basis = pt.camera_basis(width=size, height=size)
    for walls in (True, False):
        spheres = pt.scene_random(300, seed=3, with_walls=walls)
        want = first_hit_frame(size, size, spheres, basis)
        assert want[..., 6].max() > 0.0
        for variant in (None, 11, 13, 14):
            img, _ = (lab if variant == 11 else pt).render_frame(size, size, 1, spheres, basis, max_bounces=2, variant=variant)
            _check(img, want, f"HIP random300 walls={walls} variant {variant}")
