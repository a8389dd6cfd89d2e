#!/usr/bin/env python3
"""Refresh profiles/hbm_traffic.json and profiles/valu_roofline.json (read by bench.py) from a
summarised profile: update_roofline_json.py <tag> <round> <key>   e.g.  v6j r02 cfg2_xorwow_v6
Each record carries the fingerprint of the library build it was measured on (pt_build_fingerprint) and the
kernel's VGPR count; bench.py reports null instead of these counters when the loaded library differs."""
import json
import os
import sys

tag, rnd, key = sys.argv[1], sys.argv[2], sys.argv[3]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = json.load(open(os.path.join(root, "profiles", rnd, f"{tag}.json")))
src = f"profiles/{rnd}/{tag}.json"
ident = {"fingerprint": d.get("fingerprint"), "num_vgprs": d.get("num_vgprs"), "source": src}
h = d.get("hbm_bytes_per_launch")  # (absent: a profile without the TCC passes)
tp = os.path.join(root, "profiles", "hbm_traffic.json")
t = json.load(open(tp)) if os.path.exists(tp) else {}
if h:
    t[key] = dict({"hbm_bytes_per_launch": h["write"] + h["fetch_x2"], "write": h["write"], "fetch_raw": h["fetch_raw"],
                   "fetch_x2": h["fetch_x2"]}, **ident)
    json.dump(t, open(tp, "w"), indent=1)
v = d.get("valu")
if v:
    c = d["pmc_per_launch_avg"]
    vp = os.path.join(root, "profiles", "valu_roofline.json")
    r = json.load(open(vp)) if os.path.exists(vp) else {}
    samples = d.get("samples_per_launch")
    if not samples:  # a bench.py profile: the samples of one step, from the metric's "<W>x<H>x<spp>spp"
        import re
        m = re.search(r"(\d+)x(\d+)x(\d+)spp", (d.get("bench_line_under_profiler") or {}).get("metric", ""))
        samples = int(m.group(1)) * int(m.group(2)) * int(m.group(3)) if m else 2 ** 30
    r[key] = dict({"valu_insts_per_launch": v["insts_per_launch"], "kernel_ms": v["kernel_ms"], "samples_per_launch": samples,
                   "achieved_ginst_per_s": v["achieved_ginst_per_s"], "peak_ginst_per_s": v["peak_ginst_per_s"], "frac": v["frac"],
                   "flops_fp32_per_launch": (v["flops_fp32"] or 0) * 64, "flops_fp64_per_launch": (v["flops_fp64"] or 0) * 64,
                   "lane_insts_per_sample": v["insts_per_launch"] * 64 / samples,
                   "mix": v["mix"], "salu_insts": c.get("SQ_INSTS_SALU"), "branch_insts": c.get("SQ_INSTS_BRANCH"),
                   "wait_inst_any_over_wave_cycles": d.get("derived", {}).get("wait_inst_any_over_wave_cycles")}, **ident)
    json.dump(r, open(vp, "w"), indent=1)
    print(key, "valu insts %.4g" % v["insts_per_launch"], "kernel ms %.3f" % v["kernel_ms"], "frac of 1228.8 G/s: %.3f" % v["frac"])
if h:
    print(key, "hbm bytes/launch %.4g" % t[key]["hbm_bytes_per_launch"], "fingerprint", ident["fingerprint"])
