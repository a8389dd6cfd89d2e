"""bench.py end to end, including the multi-rank path the driver runs at N = 2/4/8: two ranks
under torch.distributed.run share GPU 0 (gloo + host staging stand in for RCCL, which needs one
GPU per rank), each renders its row tile with the HIP kernel, rank 0 gathers.  The gathered
frame must equal the single-process frame bit for bit, and the JSON line must keep the contract."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _run(cmd, env=None):
    e = dict(os.environ)
    e.update(env or {})
    res = subprocess.run(cmd, cwd=ROOT, env=e, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, res.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_single_and_two_ranks_agree(gpu, tmp_path):
    common = ["--steps", "2", "--warmup", "1", "--spp", "8", "--no-cpu-baseline"]
    one = _run([sys.executable, "bench.py", "--gpus", "1", "--dump", str(tmp_path / "one.npy")] + common)
    two = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                "127.0.0.1", "--master-port", "29541", "bench.py", "--gpus", "2", "--dump", str(tmp_path / "two.npy")] + common,
               env={"PT_BENCH_SHARED_GPU": "1", "PT_BENCH_BACKEND": "gloo"})
    for j, n in ((one, 1), (two, 2)):
        assert j["n_gpus"] == n and j["steps"] == 2 and j["warmup"] == 1
        assert j["unit"] == "Msamples/s" and j["higher_is_better"] is True and j["scaling"] == "strong"
        assert j["value"] > 0 and j["ms_per_step"] > 0 and j["vs_baseline"] is None
        assert set(j["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
        assert "workload" in j["config"]
    a, b = np.load(tmp_path / "one.npy"), np.load(tmp_path / "two.npy")
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))  # tiling + gather change no pixel, even on frame 3
    # the form the driver records -- `python bench.py --gpus N`, no launcher: bench.py starts its own ranks
    three = _run([sys.executable, "bench.py", "--gpus", "3", "--dump", str(tmp_path / "three.npy")] + common,
                 env={"PT_BENCH_SHARED_GPU": "1", "PT_BENCH_BACKEND": "gloo"})
    assert three["n_gpus"] == 3 and "rows/3" in three["config"]["tiling"]
    assert np.array_equal(a.view(np.uint32), np.load(tmp_path / "three.npy").view(np.uint32))
    # one process, libptcore's own multi-GPU entry (pt_mgpu_*), ranks sharing GPU 0
    nat = _run([sys.executable, "bench.py", "--gpus", "2", "--engine", "native", "--dump", str(tmp_path / "nat.npy")] + common,
               env={"PT_BENCH_SHARED_GPU": "1", "PT_FORCE_MGPU": "1"})
    assert nat["n_gpus"] == 2 and nat["config"]["engine"] == "native" and "hipMemcpyPeerAsync" in nat["config"]["tiling"]
    assert np.array_equal(a.view(np.uint32), np.load(tmp_path / "nat.npy").view(np.uint32))


def test_bench_config3_line(gpu):
    """--config cfg3 (4096 x 4096 x 64 spp, BASELINE.json configs[2]) runs on one GPU and names its workload."""
    j = _run([sys.executable, "bench.py", "--config", "cfg3", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-alt-rng"])
    assert "4096x4096" in j["metric"] and "configs[2]" in j["config"]["workload"] and j["n_gpus"] == 1
    assert 40 < j["ms_per_step"] < 200 and j["roofline"]["kernel_ms"] > 0
