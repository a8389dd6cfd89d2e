"""Known answers that do NOT come from the oracle's author's code: scenes whose per-sample result the reference's source lines
determine in closed form, evaluated here with numpy float32 in the float order of those lines, and compared BIT FOR BIT with
the oracle (CPU suite) and with the HIP kernels (-m gpu).

The scene: ONE sphere around the camera, emission E, colour rho (src/pathtrace.cu line numbers).  Every ray hits it from inside,
at every bounce, whatever the generator draws; the loop body (:155-196) then does, per bounce n,
    color += (n == 0) ? clamp(mask * emission, 0, 1) : mask * emission     (:171-174)
    mask  *= sphere.color                                                  (:175)
and after max_bounces bounces  output.color += color  (:198)  and the colour-variance update with luminance(color)  (:200).
So every sample of every pixel produces the SAME colour  c = (((t0 + t1) + t2) + ...)  with  t_n = fl(mask_n * E),
mask_0 = 1, mask_{n+1} = fl(mask_n * rho), all float32 -- independent of ray directions, random numbers and the sphere test's
arithmetic.  Hence per pixel: colour = fl(sum of spp copies of c, added one by one) / spp  (:234), colour variance exactly 0
(Welford with equal inputs: :52-58), albedo = fl(sum of spp copies of rho) / spp  (:189, :236), albedo variance exactly 0.
Depth and normal depend on the float t of the sphere test and are checked against the geometry (t = r from the centre) only
to a tolerance."""
import sys

import numpy as np
import pytest

from conftest import GOLDEN

sys.path.insert(0, GOLDEN)
from make_golden import closed_form_basis  # noqa: E402  (the camera basis as an explicit input: SURVEY 8(a) a11)

f32 = np.float32
EYE = (50.0, 52.0, 295.6)


def one_sphere(pt_or_oracle, emission, color, radius=40.0, centre=EYE):
    s = np.zeros(1, dtype=pt_or_oracle.SPHERE_DTYPE)
    s["radius"] = radius
    s["pos"] = centre
    s["emission"] = emission
    s["color"] = color
    return s


def sample_colour(emission, color, max_bounces):
    """:171-175 and :198 for one sample, channel by channel, in float32."""
    c = np.zeros(3, dtype=f32)
    mask = np.ones(3, dtype=f32)
    E = np.asarray(emission, dtype=f32)
    rho = np.asarray(color, dtype=f32)
    for n in range(max_bounces):
        me = (mask * E).astype(f32)                                                      # mask * hitObject.emission
        c = (c + (np.minimum(np.maximum(me, f32(0)), f32(1)) if n == 0 else me)).astype(f32)   # :171-174
        mask = (mask * rho).astype(f32)                                                  # :175
    return c


def pixel_mean(value, spp):
    """spp equal samples accumulated one by one (:198 / :189) and divided by (float)spp (:234-236)."""
    acc = np.zeros_like(value, dtype=f32)
    for _ in range(spp):
        acc = (acc + value).astype(f32)
    return (acc / f32(spp)).astype(f32)


CASES = [
    # emission, colour, max_bounces, spp
    ((0.25, 0.5, 0.125), (0.75, 0.5, 0.25), 5, 4),        # E < 1: no clamp
    ((0.3, 0.7, 0.9), (0.9, 0.8, 0.7), 5, 7),            # not exactly representable, odd spp
    ((4.0, 3.6, 3.2), (0.6, 0.7, 0.8), 5, 5),            # E > 1: the first bounce clamps (:172), the later ones do not
    ((1.5, 0.5, 1.0), (0.999, 0.001, 0.5), 8, 3),        # mixed clamp, 8 bounces
    ((0.1, 0.2, 0.3), (0.4, 0.5, 0.6), 1, 6),            # a single bounce
    ((0.1, 0.2, 0.3), (0.4, 0.5, 0.6), 2, 2),
    ((0.1, 0.2, 0.3), (0.4, 0.5, 0.6), 3, 16),
    ((0.1, 0.2, 0.3), (0.4, 0.5, 0.6), 4, 1),            # spp 1: no jitter (:222), variance 0 by definition (:61)
    ((0.1, 0.2, 0.3), (0.4, 0.5, 0.6), 6, 9),
    ((0.1, 0.2, 0.3), (0.4, 0.5, 0.6), 7, 2),
    ((0.0, 0.0, 0.0), (0.75, 0.75, 0.75), 5, 4),         # black: colour exactly 0
]


def check_frame(img, emission, color, max_bounces, spp, radius):
    want_c = pixel_mean(sample_colour(emission, color, max_bounces), spp)
    want_a = pixel_mean(np.asarray(color, dtype=f32), spp)
    h, w = img.shape[:2]
    assert np.array_equal(img[..., 0:3].view(np.uint32), np.broadcast_to(want_c, (h, w, 3)).view(np.uint32)), (img[0, 0, 0:3], want_c)
    assert np.array_equal(img[..., 6:9].view(np.uint32), np.broadcast_to(want_a, (h, w, 3)).view(np.uint32)), (img[0, 0, 6:9], want_a)
    assert np.all(img[..., 10] == 0.0) and np.all(img[..., 12] == 0.0)  # colour and albedo variance: equal samples
    # geometry (not closed-form in float): from the centre every hit is at distance r, i.e. at ray parameter r / |d| with the
    # unnormalised primary direction d (:229) between the basis' axis length and its corner length
    b = np.asarray(closed_form_basis(w, h), dtype=np.float64).reshape(4, 3)
    d_max = np.linalg.norm(b, axis=1).max() * (1 + 1.0 / min(w, h))  # (the jitter reaches half a pixel beyond the corners)
    d_min = np.linalg.norm(b.mean(axis=0)) * (1 - 1e-5)
    assert np.all(img[..., 9] >= radius / d_max) and np.all(img[..., 9] <= radius / d_min)
    nrm = np.linalg.norm(img[..., 3:6].astype(np.float64), axis=-1)
    assert np.all(nrm <= 1.0 + 1e-5) and np.all(nrm > 0.99)  # an average of unit vectors a milliradian apart


@pytest.mark.parametrize("emission,color,max_bounces,spp", CASES)
@pytest.mark.parametrize("rng", ["xorwow", "philox"])
def test_oracle_reproduces_the_closed_form(oracle, emission, color, max_bounces, spp, rng):
    radius = 40.0
    mode = oracle.RNG_XORWOW if rng == "xorwow" else oracle.RNG_PHILOX
    img = oracle.render(24, 16, spp, one_sphere(oracle, emission, color, radius), closed_form_basis(24, 16), max_bounces=max_bounces, rng_mode=mode, threads=2)
    check_frame(img, emission, color, max_bounces, spp, radius)


@pytest.mark.gpu
@pytest.mark.parametrize("emission,color,max_bounces,spp", CASES)
def test_hip_kernels_reproduce_the_closed_form(pt, gpu, emission, color, max_bounces, spp):
    radius = 40.0
    scene = one_sphere(pt, emission, color, radius)
    for rng in (pt.RNG_XORWOW, pt.RNG_PHILOX):
        for variant in (None, 0, 6, 8, 10):  # automatic, literal, screened, four lanes per pixel, regeneration
            img, _ = pt.render_frame(96, 32, spp, scene, closed_form_basis(96, 32), max_bounces=max_bounces, rng_mode=rng, variant=variant)
            check_frame(img, emission, color, max_bounces, spp, radius)


@pytest.mark.gpu
def test_hip_grid_kernel_reproduces_the_closed_form(pt, gpu):
    """The many-sphere kernel (variant 13: grid, pooled tests, per-pixel primary lists) on the same construction: the enclosing
    sphere plus 200 small, brightly emitting decoys far outside it (never the nearest hit: the closed form is unchanged), so that
    a grid exists and every pixel's primary-ray list is non-trivial."""
    rs = np.random.default_rng(5)
    emission, color, radius = (0.3, 0.7, 0.9), (0.9, 0.8, 0.7), 40.0
    scene = np.zeros(201, dtype=pt.SPHERE_DTYPE)
    scene[0] = one_sphere(pt, emission, color, radius)[0]
    for k in range(1, 201):  # decoys on a shell of radius 200-260 around the camera: outside the enclosing sphere
        v = rs.normal(size=3)
        v /= np.linalg.norm(v)
        scene[k]["pos"] = np.asarray(EYE) + v * rs.uniform(200.0, 260.0)
        scene[k]["radius"] = rs.uniform(1.0, 4.0)
        scene[k]["color"] = (0.5, 0.5, 0.5)
        scene[k]["emission"] = (9.0, 9.0, 9.0)  # would show at once if a ray ever got there
    for rng in (pt.RNG_XORWOW, pt.RNG_PHILOX):
        for mb, spp in ((5, 8), (3, 5), (8, 4)):
            img, _ = pt.render_frame(128, 64, spp, scene, closed_form_basis(128, 64), max_bounces=mb, rng_mode=rng, variant=13)
            check_frame(img, emission, color, mb, spp, radius)
