"""The TOLERANCED fast mode (pt_renderer_opts.fast_math = 1, csrc/pt_fast.hip): same algorithm and generator streams
as the bit-exact kernels, but FMA contraction, an FP32-only cancellation-free sphere test and hardware
rsq/sin/cos/rcp.  It is reported beside the exact kernels, so its acceptance is statistical, and the tolerances are
written here:

  T1  1 spp (no jitter, first hit only): albedo identical in >= 99.8 % of the pixels (a primary ray within rounding of a
      sphere's edge may pick the other surface), normals within 2e-4 in 99.9 % of them (2e-3 at grazing hits), depth
      within 5e-4 relative;
  T2  per-channel image means at 256 x 256 x 256 spp agree with the exact kernel's within 4 standard errors of the
      Monte-Carlo mean (standard error from the frame's own variance channels);
  T3  at equal seeds and 64 spp, at most 3 % of the pixels differ by more than 1e-4 in colour (measured 1.6 %) and the
      median difference is 0: the integrand is chaotic -- one differing rounding sends a path to another surface -- and
      the reference's own source compiled with and without contraction already differs in ~1 % of the pixels
      (SURVEY.md fact 5).  The share grows with spp (one divergent path in a pixel is enough: 6.6 % at 256 spp, 23 % at
      1024 spp) while each divergent path weighs 1/spp;
  T4  the row-tile independence and the generator-state persistence of the exact kernels hold bit for bit
      (fast mode against fast mode);
  T5  the distance from the ORACLE (the reference's CPU restatement) per channel at 256 x 256 x {1, 64} spp: the table the
      tolerances in include/ptcore.h quote (round 4)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _pair(pt, size, spp, **kw):
    basis = pt.camera_basis(width=size, height=size)
    exact, ms_e = pt.render_frame(size, size, spp, basis=basis, **kw)
    fast, ms_f = pt.render_frame(size, size, spp, basis=basis, fast_math=True, **kw)
    return exact, fast, ms_e, ms_f


@pytest.mark.parametrize("rng", [0, 1], ids=["xorwow", "philox"])
def test_t1_first_hit_features_at_one_sample(pt, gpu, rng):
    exact, fast, _, _ = _pair(pt, 256, 1, rng_mode=rng)
    assert np.isfinite(fast).all()
    same_albedo = np.all(exact[..., 6:9] == fast[..., 6:9], axis=-1)
    assert (~same_albedo).mean() <= 2e-3, f"{(~same_albedo).sum()} pixels see another surface"
    ok = same_albedo
    nd = np.abs(exact[ok][:, 3:6] - fast[ok][:, 3:6]).max(axis=-1)               # unit normals
    # grazing hits on the small spheres amplify the FP32 discriminant's rounding (measured: median 4e-9, p99.9 6e-5, max 5e-4)
    assert np.quantile(nd, 0.999) <= 2e-4 and nd.max() <= 2e-3
    rel_depth = np.abs(exact[ok][:, 9] - fast[ok][:, 9]) / exact[ok][:, 9]
    # depth = t along the un-normalised primary direction (|d| ~ 0.02), the reference's own float c loses ~1e-4 of it
    assert np.quantile(rel_depth, 0.999) <= 5e-4 and np.median(rel_depth) <= 2e-5
    assert np.all(fast[..., 10:] == 0)                                             # variances of a single sample


def test_t2_image_means_within_monte_carlo_error(pt, gpu):
    size, spp = 256, 256
    exact, fast, ms_e, ms_f = _pair(pt, size, spp)
    npx = size * size
    for name, ch, var_ch in (("colour", (0, 1, 2), 10), ("normal", (3, 4, 5), 11), ("albedo", (6, 7, 8), 12), ("depth", (9,), 13)):
        # standard error of the image mean of one feature component <= sqrt(sum of per-pixel variance / spp) / pixels;
        # the variance channels hold the variance of the LUMINANCE-reduced feature, a fair scale for each component
        se = np.sqrt(exact[..., var_ch].astype(np.float64).sum() / spp) / npx
        for c in ch:
            a, b = exact[..., c].mean(dtype=np.float64), fast[..., c].mean(dtype=np.float64)
            assert abs(a - b) <= 4.0 * np.sqrt(2.0) * se + 2e-6 * abs(a), f"{name}[{c}]: exact {a:.7g} fast {b:.7g} se {se:.3g}"
    # the variance estimates themselves agree on average (2 %)
    for var_ch in (10, 11, 12, 13):
        a, b = exact[..., var_ch].mean(dtype=np.float64), fast[..., var_ch].mean(dtype=np.float64)
        assert abs(a - b) <= 0.02 * abs(a) + 1e-9, (var_ch, a, b)
    print(f"256x256x256spp: exact {ms_e:.2f} ms, fast {ms_f:.2f} ms")


def test_t3_pixelwise_divergence_is_bounded(pt, gpu):
    exact, fast, _, _ = _pair(pt, 256, 64)
    diff = np.abs(exact[..., :3] - fast[..., :3]).max(axis=-1)
    share = (diff > 1e-4).mean()
    print(f"64 spp: {100 * share:.2f} % of pixels differ by more than 1e-4 in colour, median {np.median(diff):.2e}, max {diff.max():.3g}")
    assert share <= 0.03 and np.median(diff) <= 1e-6
    # first-hit features are averaged over jittered samples, far less chaotic: 99 % of pixels within 1e-4 (normals)
    nd = np.abs(exact[..., 3:6] - fast[..., 3:6]).max(axis=-1)
    assert (nd > 1e-4).mean() <= 0.01


_CHANNELS = [("colour", (0, 1, 2), False), ("normal", (3, 4, 5), False), ("albedo", (6, 7, 8), False), ("depth", (9,), True),
             ("colourVar", (10,), False), ("normalVar", (11,), False), ("albedoVar", (12,), False), ("depthVar", (13,), True)]


def _distance_table(ref, fast):
    """Per channel group: L-infinity, 99.9th percentile, median and share of pixels beyond 1e-4 (depth and its variance
    RELATIVE: depth is t along the un-normalised primary direction, about 1e4)."""
    rows = {}
    for name, ch, rel in _CHANNELS:
        a, b = ref[..., ch].astype(np.float64), fast[..., ch].astype(np.float64)
        d = np.abs(a - b)
        if rel:
            d = d / np.maximum(np.abs(a), 1e-30)
        d = d.max(axis=-1)
        rows[name] = {"relative": rel, "linf": float(d.max()), "p999": float(np.quantile(d, 0.999)), "median": float(np.median(d)),
                      "share_gt_1e-4": float((d > 1e-4).mean())}
    return rows


@pytest.mark.parametrize("rng", [0, 1], ids=["xorwow", "philox"])
def test_t5_distance_from_the_oracle(pt, oracle, gpu, rng):
    """The north star states its tolerance against the REFERENCE ("per-pixel L-inf <= 1e-4 at fixed seed"), so the fast mode is
    measured against the reference's oracle, not only against this repository's other kernel: 256 x 256 x {1, 64} spp, equal
    seeds.  What holds (and is quoted in include/ptcore.h):
      1 spp  -- albedo identical in >= 99.8 % of the pixels; where it is: normals p99.9 <= 2e-4, depth p99.9 <= 5e-4 relative;
      64 spp -- colour beyond 1e-4 in <= 3 % of the pixels (median 0), normals in <= 1 %, albedo in <= 1 %, depth p99.9 <= 5e-3
                relative: one ray in 64 that rounds onto another surface moves a pixel by up to 1/64 of the feature's range.
    The L-inf of the north star is NOT met by any implementation that rounds differently from the reference -- the reference's
    own source compiled with and without FMA contraction is 0.06 apart in colour at 64 spp -- which is why
    the headline stays on the bit-exact kernels.  PT_FAST_TABLE_OUT=<file>: the full table as JSON (profiles/r04/fast_vs_oracle.json)."""
    import json
    import os

    size = 256
    basis = pt.camera_basis(width=size, height=size)
    record = {"fingerprint": pt.build_fingerprint(), "size": size, "rng": ["xorwow", "philox"][rng], "cases": []}
    for spp in (1, 64):
        ref = oracle.render(size, size, spp, spheres=pt.scene_cornell(), basis=basis, rng_mode=rng)
        fast, _ = pt.render_frame(size, size, spp, basis=basis, rng_mode=rng, fast_math=True)
        assert np.isfinite(fast).all()
        same = np.all(ref[..., 6:9] == fast[..., 6:9], axis=-1)
        t = _distance_table(ref, fast)
        record["cases"].append({"spp": spp, "albedo_identical_share": float(same.mean()), "channels": t})
        print(f"{record['rng']} {spp} spp: albedo identical {same.mean():.4f}; " +
              "; ".join(f"{k} Linf {v['linf']:.3g} p99.9 {v['p999']:.3g} >1e-4 {100 * v['share_gt_1e-4']:.3f}%" for k, v in t.items()))
        if spp == 1:
            assert same.mean() >= 0.998
            ts = _distance_table(ref[same][None], fast[same][None])
            assert ts["normal"]["p999"] <= 2e-4 and ts["depth"]["p999"] <= 5e-4
            assert all(t[k]["linf"] == 0.0 for k in ("colourVar", "normalVar", "albedoVar", "depthVar"))
        else:
            assert t["colour"]["share_gt_1e-4"] <= 0.03 and t["colour"]["median"] <= 1e-6
            assert t["normal"]["share_gt_1e-4"] <= 0.01 and t["albedo"]["share_gt_1e-4"] <= 0.01
            assert t["depth"]["p999"] <= 5e-3
    out = os.environ.get("PT_FAST_TABLE_OUT")
    if out:
        prev = json.load(open(out)) if os.path.exists(out) else []
        json.dump([r for r in prev if r.get("rng") != record["rng"]] + [record], open(out, "w"), indent=1)


def test_t4_tiles_and_generator_state(pt, gpu):
    size, spp = 96, 4
    basis = pt.camera_basis(width=size, height=size)
    full, _ = pt.render_frame(size, size, spp, basis=basis, fast_math=True)
    for b, e in ((0, 13), (13, 70), (70, 96)):
        tile, _ = pt.render_frame(size, size, spp, basis=basis, fast_math=True, row_begin=b, row_end=e)
        assert np.array_equal(tile.view(np.uint32), full[b:e].view(np.uint32))
    # the generator stream is the exact kernels' own: after one frame of a closed scene both have drawn 12 x spp numbers
    re_, rf = pt.Renderer(size, size, spp), pt.Renderer(size, size, spp, fast_math=True)
    assert rf.kernel_info(9)["variant"] == pt.VARIANT_FAST
    d_scene, n = pt.upload_scene(pt.scene_cornell())
    d_out = pt.DeviceBuffer(size * size * 56)
    re_.render(d_out.ptr, d_scene.ptr, n, basis)
    rf.render(d_out.ptr, d_scene.ptr, n, basis)
    # ... except in the very few pixels where a rounding let a fast-mode ray slip out between two wall spheres (a path
    # that ends early draws less): bounded at one pixel in a thousand
    differ = np.any(re_.get_rng_state() != rf.get_rng_state(), axis=1).mean()
    print(f"generator state differs in {100 * differ:.3f} % of the pixels")
    assert differ <= 1e-3
    re_.destroy()
    rf.destroy()
    with pytest.raises(pt.PtError):
        pt.Renderer(8, 8, 1, fast_math=True, variant=6)


def test_many_spheres_and_open_scene(pt, gpu):
    """Generic build (run-time sphere count and bounce cap; > 64 spheres are read in place): statistics against the
    exact kernel on a 300-sphere scene, closed and open."""
    size, spp = 128, 64
    basis = pt.camera_basis(width=size, height=size)
    for walls in (True, False):
        scene = pt.scene_random(300, seed=9, with_walls=walls)
        exact, _ = pt.render_frame(size, size, spp, spheres=scene, basis=basis, max_bounces=4)
        fast, _ = pt.render_frame(size, size, spp, spheres=scene, basis=basis, max_bounces=4, fast_math=True)
        assert np.isfinite(fast).all()
        for c in (0, 1, 2, 6, 7, 8):
            a, b = exact[..., c].mean(dtype=np.float64), fast[..., c].mean(dtype=np.float64)
            assert abs(a - b) <= 0.01 * abs(a) + 2e-4, (walls, c, a, b)
