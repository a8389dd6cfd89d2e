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


def test_bench_other_configuration_lines(gpu):
    """--config cfg4 / cfg4open / cfg5 (BASELINE.json configs[3], configs[4]) print the same line format and name their workload."""
    for key, needle in (("cfg4", "configs[3], closed"), ("cfg4open", "configs[3], open"), ("cfg5", "configs[4]")):
        j = _run([sys.executable, "bench.py", "--config", key, "--steps", "2", "--warmup", "1", "--no-cpu-baseline"])
        assert needle in j["config"]["workload"] and j["n_gpus"] == 1 and j["value"] > 0
        assert j["roofline"]["kernel_ms"] > 0
        # counters come from profiles/*.json only when they were taken on THIS build (round 5: every configuration has a record at its
        # own size); a stale or missing profile leaves null and says why -- never numbers of another build
        assert (j["roofline"]["traffic"] is None) == (j["profile_stale"] is not None)
        assert j["roofline"]["traffic"] is None or j["roofline"]["traffic"] > 56 * 512 * 512
        assert (j["counter_based_rng"] is None) == (key != "cfg5") and (j["fast_mode"] is None) == (key != "cfg5")
    assert "8 bounces" in j["metric"] and "512x512" in j["metric"]


def test_bench_rank_failure_kills_the_job(gpu, tmp_path):
    """`python bench.py --gpus 3` with one rank dying before the first frame: the launcher exits non-zero promptly and leaves
    no rank behind (the survivors would otherwise wait in the gather forever)."""
    import time
    env = dict(os.environ, PT_BENCH_SHARED_GPU="1", PT_BENCH_BACKEND="gloo", PT_BENCH_TEST_DIE_RANK="1", PT_BENCH_TEST_PIDDIR=str(tmp_path))
    t0 = time.perf_counter()
    res = subprocess.run([sys.executable, "bench.py", "--gpus", "3", "--steps", "2", "--warmup", "1", "--spp", "8", "--no-cpu-baseline"],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 17, (res.returncode, res.stderr[-1500:])
    assert time.perf_counter() - t0 < 240
    assert not [l for l in res.stdout.splitlines() if l.startswith('{"metric"')]  # no bench line from a broken job
    pids = [int(f.name) for f in tmp_path.iterdir()]
    assert len(pids) == 3
    time.sleep(0.5)
    for pid in pids:
        with pytest.raises(ProcessLookupError):
            os.kill(pid, 0)
