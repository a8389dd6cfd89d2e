"""The automatic kernel choice for small scenes is a COST MODEL (csrc/pt_capi.hip, "which kernel for a small scene"): predicted
kernel time of variants 6 / 8 / 9 from three measured constants per kernel.  It must reproduce the measured optimum of this
round's sweeps (profiles/r04/: row tiles of the headline frame at 1024 spp, and short frames of 256^2 ... 640^2 at 4 ... 256 spp,
both generators): the variant it picks may be at most 2 % (long frames) / 3 % (short frames) slower than the fastest measured
one at EVERY point.  Host arithmetic only -- no GPU (pt_debug_policy_ms, lab library)."""
import ast
import os
import re

import pytest

from conftest import ROOT

R04 = os.path.join(ROOT, "profiles", "r04")


def _rows(path):
    for line in open(path):
        m = re.search(r"\{.*\}", line)
        if m:
            yield line[:m.start()], ast.literal_eval(m.group(0))


def _gap(lab, rng, w, spp, bounces, times):
    pick = lab.policy_choice(rng, w, spp, bounces)
    return pick, times[f"v{pick}"] / min(times[k] for k in ("v6", "v8", "v9")) - 1.0


def test_model_reproduces_the_measured_optimum_on_row_tiles(lab):
    n = 0
    for head, t in _rows(os.path.join(R04, "tile_policy_sweep.txt")):
        m = re.match(r"rng (\d) rows\s+(\d+) \(\s*([\d.]+)", head)
        rng, w = int(m.group(1)), float(m.group(3))
        pick, gap = _gap(lab, rng, w, 1024, 5, t)
        assert gap <= 0.02, f"rng {rng} w {w}: the model picks variant {pick}, {100 * gap:.1f} % slower than the best of {t}"
        n += 1
    assert n == 24


@pytest.mark.parametrize("name,rng", [("xorwow", 0), ("philox", 1)])
def test_model_reproduces_the_measured_optimum_on_short_frames(lab, name, rng):
    n = 0
    for head, t in _rows(os.path.join(R04, f"short_frames_{name}.txt")):
        m = re.search(r"\(w=([\d.]+)\) x (\d+) spp x (\d) bounces", head)
        w, spp, bounces = float(m.group(1)), int(m.group(2)), int(m.group(3))
        pick, gap = _gap(lab, rng, w, spp, bounces, t)
        assert gap <= 0.03, f"{name} w {w} spp {spp}: the model picks variant {pick}, {100 * gap:.1f} % slower than the best of {t}"
        n += 1
    assert n == 35


def test_model_shape(lab):
    """Sanity of the three regimes: a lone wave is latency-bound (more waves of the same kernel cost nothing until the SIMD is
    issue-bound), time grows linearly from there, and a frame too short for sample chunking pays whole rounds."""
    one, two, eight, sixteen = (lab.policy_ms(0, 6, w, 1024) for w in (1, 2, 8, 16))
    assert two < 1.2 * one and 1.9 < sixteen / eight < 2.1
    assert lab.policy_ms(0, 6, 5.06, 64) > 1.3 * lab.policy_ms(0, 6, 5.0, 64)  # a sixth wave per SIMD starts a second round
    assert lab.policy_choice(0, 1.0, 1024) == 8 and lab.policy_choice(0, 16.0, 1024) == 6 and lab.policy_choice(0, 2.0, 1024) == 9
    assert lab.policy_choice(0, 2.0, 1024, with9=False) == 8
    # the C++ choice is the argmin of the C++ model (pt_debug_policy_choice against pt_debug_policy_ms)
    for rng in (0, 1):
        for w in (0.5, 1.0, 2.0, 3.5, 8.0, 16.0):
            for spp in (4, 64, 1024):
                ms = {v: lab.policy_ms(rng, v, w, spp) for v in (6, 8, 9)}
                assert ms[lab.policy_choice(rng, w, spp)] == min(ms.values()), (rng, w, spp, ms)
