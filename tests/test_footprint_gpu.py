"""The pixel-footprint exclusion of the bounce-0 screen (csrc/pt_footprint.h; reference-configuration builds of variants 6 and 8,
spp >= 8): the Cornell box with perturbed spheres and cameras placed where the exclusion's criteria are closest to their limits --
the eye next to a sphere (c near 0), hugging a wall or in a corner (grazing walls), spheres poking through walls, nested and
duplicated spheres, a scaled scene, a non-power-of-two image -- at resolutions where the exclusion is active, whole frames (or a
128-row tile of the larger ones) against the oracle bit for bit.  tools/footprint_soak.py is the long form of this test
(profiles/r02/footprint_soak_600.json), including two deliberately unsound mutants that it catches."""
import numpy as np
import pytest

from test_parity_gpu import assert_bit_exact

pytestmark = pytest.mark.gpu


def _case(pt, seed):
    rng = np.random.default_rng(91000 + seed)
    sc = pt.scene_cornell().copy()
    k = seed % 8
    scale = 1.0
    if k >= 1:
        sc["pos"][6:] += rng.normal(0, 8.0, size=(3, 3)).astype(np.float32)
        sc["radius"][6:8] *= rng.uniform(0.3, 2.5, 2).astype(np.float32)
    if k == 2:
        sc["pos"][:6] += rng.normal(0, 1.5, size=(6, 3)).astype(np.float32)
    if k == 3:
        sc["pos"][6] = (rng.uniform(0, 8), rng.uniform(0, 20), rng.uniform(20, 150))      # through a wall / the floor
    if k == 4:
        sc["pos"][7], sc["radius"][7] = sc["pos"][6], sc["radius"][6]                       # duplicate: first index wins
    if k == 5:
        sc["radius"][6] = rng.uniform(60, 400)                                              # eye inside a non-wall sphere
    if k == 6:
        scale = float(rng.choice([0.01, 100.0]))
        sc["pos"] *= np.float32(scale)
        sc["radius"] *= np.float32(scale)
    size = int(rng.choice([256, 320, 500, 512, 1024]))
    if k == 7:
        eye = tuple(rng.choice([[1.2, 40, 100], [98.9, 5, 20], [50, 81.3, 150], [2, 1, 2], [50, 40, 598]]) + rng.normal(0, 0.05, 3))
        yaw = float(rng.uniform(-180, 180))
    elif k == 1:
        eye = tuple((sc["pos"][6] + rng.normal(0, 1, 3) * (sc["radius"][6] * rng.choice([1.001, 1.02, 1.3]) + rng.choice([0.0, 5.0]))).astype(float))
        yaw = float(rng.uniform(-130, -50))
    else:
        eye = tuple(np.array(rng.uniform([10, 10, 120], [90, 70, 320])) * scale)
        yaw = float(rng.uniform(-130, -50))
    basis = pt.camera_basis(eye, yaw, float(rng.uniform(-40, 40)), size, size)
    rb = 0 if size <= 512 else int(rng.integers(0, size - 128))
    re_ = size if size <= 512 else rb + 128
    return sc, eye, basis, size, rb, re_, int(rng.choice([8, 9, 16]))


@pytest.mark.parametrize("seed", range(16))
def test_footprint_exclusion_changes_no_pixel(pt, oracle, gpu, seed):
    sc, eye, basis, size, rb, re_, spp = _case(pt, seed)
    mode = seed % 2
    ref = oracle.render(size, size, spp, spheres=sc, basis=basis, eye=eye, rng_mode=mode, row_begin=rb, row_end=re_)
    for v in (6, 8, None):
        img, _ = pt.render_frame(size, size, spp, spheres=sc, basis=basis, eye=eye, rng_mode=mode, variant=v, row_begin=rb, row_end=re_)
        assert_bit_exact(img, ref, f"footprint case {seed} ({size}px, {spp} spp, rng {mode}) variant {v}")
