"""bench.py's host-side pieces that need no GPU: the configuration table and the CPU-baseline leg (the oracle timed on a
bounded sample of the same workload) for the smallest configuration, whose whole frame is less CPU work than the budget."""
import sys

from conftest import ROOT

sys.path.insert(0, ROOT)


def test_configs_name_baseline_json_entries():
    import bench

    assert set(bench.CONFIGS) == {"cfg2", "cfg3", "cfg4", "cfg4open", "cfg5"}
    assert bench.CONFIGS["cfg2"]["spp"] == 1024 and bench.CONFIGS["cfg5"]["max_bounces"] == 8
    for key, c in bench.CONFIGS.items():
        assert "BASELINE.json configs[" in c["name"], key


def test_cpu_baseline_of_a_frame_smaller_than_the_budget(pt, oracle):
    import bench

    cfg = dict(bench.CONFIGS["cfg5"], width=64, height=64)  # the same shape, smaller: the band must stay inside the frame
    spheres, label = bench.scene_of(pt, cfg)
    out = bench.cpu_baseline(oracle, cfg, spheres, pt.camera_basis(width=64, height=64), 0)
    assert out["kind"] == "port" and out["value"] > 0 and out["cores"] >= 1
    assert "rows 0..63 of the 64x64 frame" in out["sample"] and "rendered" in out["sample"]
    spheres4, label4 = bench.scene_of(pt, bench.CONFIGS["cfg4open"])
    assert len(spheres4) == 1000 and "without walls" in label4
