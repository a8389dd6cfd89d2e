"""Degenerate scenes (NaN / infinite / huge / denormal sphere data, duplicated spheres, the eye exactly on a surface looking
along a tangent): what the exactness arguments of EXACTNESS.md say about NaN propagation, rejected candidates and
doubted estimates, checked against the oracle bit for bit on every kernel family (tools/degenerate_soak.py)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_degenerate_scenes_match_the_oracle(gpu):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "degenerate_soak.py"), "210", "7000"], capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    rec = json.loads(out.stdout.strip().splitlines()[-1])
    assert rec["cases"] == 210 and rec["n_mismatching"] == 0 and rec["floats_compared"] > 5_000_000
