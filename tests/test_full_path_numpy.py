"""Whole frames -- all 14 channels, several samples per pixel, the reference's XORWOW stream -- from the SECOND restatement of the
reference's path (tests/numpy_restatement.py: vectorised numpy written from src/pathtrace.cu:39-257, no code shared with
oracle/pt_oracle.c), compared BIT FOR BIT with the C oracle (CPU suite) and with the HIP kernels (-m gpu).

Two restatements in two languages that agree on every bit of every channel agree on every reading of the reference that can change
a bit: which sub-expressions are double (:68, :75, :80-81, :134), the float product (b*b) inside the double discriminant, `t`
persisting across the sphere loop (:95), strict compares and first-index-wins (:99), clamp on the first bounce only (:171-174),
a miss adding colour without a colour-variance update (:157-161), the int divisor of the Welford mean (:55), the draw order
jitter x, jitter y, then azimuth, elevation per bounce (:223-224, :131 + contract C5), the generator state of a pixel running on
from sample to sample, and the final divisions (:234-237, :63).  sin / cos (:135) are contract C4's polynomial in both (the numpy
side calls the oracle's pto_sincos: explicit fmaf has no numpy counterpart), the XORWOW constants are contract C8's in both."""
import numpy as np
import pytest

import numpy_restatement as NR
from test_first_hit_numpy import _scenes


def _sincos_of(oracle):
    both = np.frompyfunc(oracle.sincos, 1, 2)

    def sincos(x):
        s, c = both(x)
        return s.astype(np.float32), c.astype(np.float32)

    return sincos


def _check_all(img, want, what):
    got = np.ascontiguousarray(img).view(np.uint32)
    exp = np.ascontiguousarray(want).view(np.uint32)
    bad = np.argwhere(got != exp)
    assert bad.size == 0, f"{what}: {len(bad)} of {got.size} floats differ, first at (row, col, channel) {bad[0]}: " \
                          f"got {img[tuple(bad[0])]!r} want {want[tuple(bad[0])]!r}; channels hit {sorted(set(bad[:, 2]))}"


CASES = [  # scene, size, spp, max_bounces
    ("cornell", 64, 4, 5),           # the reference's configuration
    ("cornell", 48, 1, 5),           # one sample: no jitter draws (:222)
    ("cornell", 40, 3, 8),           # not a power of two, odd spp, BASELINE config 5's bounce cap
    ("random_open", 64, 5, 5),       # paths that leave the scene at every depth
    ("random_enclosed", 48, 4, 3),
    ("random_open", 32, 16, 1),      # a single bounce, many samples: the Welford recurrences
    ("cornell", 256, 16, 5),         # 5.2 M rays: large enough that one-ulp changes of a direction show (next test)
]


@pytest.mark.parametrize("scene,size,spp,max_bounces", CASES)
def test_oracle_frames_equal_the_numpy_restatement(oracle, scene, size, spp, max_bounces):
    spheres = _scenes(oracle)[scene]
    basis = oracle.camera_basis(w=size, h=size)
    want = NR.render_frame(size, size, spp, spheres, basis, _sincos_of(oracle), max_bounces=max_bounces)
    img = oracle.render(size, size, spp, spheres, basis, max_bounces=max_bounces, threads=4)
    assert np.isfinite(want).all() and want[..., 9].max() > 0.0
    _check_all(img, want, f"oracle {scene} {size}x{size}x{spp} b{max_bounces}")


def test_the_comparison_sees_one_ulp_changes_of_the_bounce_directions(oracle):
    """The colour of a path is a function of WHICH spheres it hits, so a last-bit change of a bounce direction shows only where it
    moves a ray across a silhouette.  At 128 x 128 x 16 spp it does: sin / cos from libm (correctly rounded; it differs from contract
    C4's polynomial in the last bit of a quarter of the arguments) instead of the contract's changes floats of the frame that the
    restatement otherwise reproduces bit for bit."""
    spheres = oracle.scene_cornell()
    basis = oracle.camera_basis(w=128, h=128)
    img = oracle.render(128, 128, 16, spheres, basis, threads=4)

    def libm(x):
        return np.sin(x.astype(np.float64)).astype(np.float32), np.cos(x.astype(np.float64)).astype(np.float32)

    _check_all(img, NR.render_frame(128, 128, 16, spheres, basis, _sincos_of(oracle)), "oracle cornell 128x128x16")
    other = NR.render_frame(128, 128, 16, spheres, basis, libm)
    assert np.count_nonzero(other.view(np.uint32) != img.view(np.uint32)) > 0


def test_numpy_xorwow_equals_the_survey_vectors(oracle):
    """the first three uniforms of seeds 0, 1, 65535 (tests/golden/survey_kats.json)."""
    import json
    import os

    from conftest import GOLDEN

    kats = json.load(open(os.path.join(GOLDEN, "survey_kats.json")))["xorwow_first3_uniforms"]
    seeds = np.asarray([int(k) for k in kats], dtype=np.uint64)
    g = NR.Xorwow(seeds)
    every = np.ones(len(seeds), dtype=bool)
    got = np.stack([g.uniform(every) for _ in range(3)], axis=1)
    np.testing.assert_allclose(got, [kats[k] for k in kats], rtol=2e-7)


@pytest.mark.gpu
@pytest.mark.parametrize("scene,size,spp,max_bounces", CASES)
def test_hip_frames_equal_the_numpy_restatement(pt, oracle, gpu, scene, size, spp, max_bounces):
    spheres = _scenes(pt)[scene]
    basis = pt.camera_basis(width=size, height=size)
    want = NR.render_frame(size, size, spp, spheres, basis, _sincos_of(oracle), max_bounces=max_bounces)
    for variant in (None, 0, 6, 8, 10):  # automatic, literal, screened, four lanes per pixel, regeneration
        img, _ = pt.render_frame(size, size, spp, spheres, basis, max_bounces=max_bounces, variant=variant)
        _check_all(img, want, f"HIP {scene} {size}x{size}x{spp} b{max_bounces} variant {variant}")


@pytest.mark.gpu
def test_hip_grid_kernels_equal_the_numpy_restatement(pt, lab, oracle, gpu):
    """The many-sphere kernels (grid walk -- variant 11 lives in the lab library; pooled tests, per-pixel primary lists at spp >= 4, last-bounce shortcut)."""
    size = 64
    basis = pt.camera_basis(width=size, height=size)
    for walls in (True, False):
        spheres = pt.scene_random(150, seed=5, with_walls=walls)
        want = NR.render_frame(size, size, 4, spheres, basis, _sincos_of(oracle), max_bounces=5)
        for variant in (None, 11, 13, 14):
            img, _ = (lab if variant == 11 else pt).render_frame(size, size, 4, spheres, basis, max_bounces=5, variant=variant)
            _check_all(img, want, f"HIP random150 walls={walls} variant {variant}")
