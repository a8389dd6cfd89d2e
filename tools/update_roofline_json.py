#!/usr/bin/env python3
"""Refresh profiles/hbm_traffic.json and profiles/valu_roofline.json (read by bench.py) from a
summarised profile: update_roofline_json.py <tag> <round> <key>   e.g.  v6c_xorwow r01 xorwow_v6"""
import json
import os
import sys

tag, rnd, key = sys.argv[1], sys.argv[2], sys.argv[3]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = json.load(open(os.path.join(root, "profiles", rnd, f"{tag}.json")))
src = f"profiles/{rnd}/{tag}.json"
h = d["hbm_bytes_per_launch"]
tp = os.path.join(root, "profiles", "hbm_traffic.json")
t = json.load(open(tp)) if os.path.exists(tp) else {}
t[key] = {"hbm_bytes_per_launch": h["write"] + h["fetch_x2"], "write": h["write"], "fetch_raw": h["fetch_raw"],
          "fetch_x2": h["fetch_x2"], "source": src}
json.dump(t, open(tp, "w"), indent=1)
v = d.get("valu")
if v:
    c = d["pmc_per_launch_avg"]
    t_s = v["kernel_ms"] / 1e3
    peak = 1024 / 1.125e-9
    vp = os.path.join(root, "profiles", "valu_roofline.json")
    r = json.load(open(vp)) if os.path.exists(vp) else {}
    r[key] = {"valu_insts_per_launch": v["insts_per_launch"], "kernel_ms": v["kernel_ms"],
              "achieved_ginst_per_s": v["insts_per_launch"] / t_s / 1e9, "peak_ginst_per_s": peak / 1e9,
              "frac": v["insts_per_launch"] / t_s / peak, "modelled_issue_bound_ms": v["issue_bound_ms"],
              "fp32_tflops": v["flops_fp32"] * 64 / t_s / 1e12, "fp64_tflops": v["flops_fp64"] * 64 / t_s / 1e12,
              "flop_per_sample_fp32": v["flops_fp32"] * 64 / 2 ** 30, "flop_per_sample_fp64": v["flops_fp64"] * 64 / 2 ** 30,
              "mix": v["mix"], "salu_insts": c.get("SQ_INSTS_SALU"), "branch_insts": c.get("SQ_INSTS_BRANCH"), "source": src}
    json.dump(r, open(vp, "w"), indent=1)
    print(key, "valu insts %.4g" % v["insts_per_launch"], "kernel ms %.3f" % v["kernel_ms"], "frac %.3f" % r[key]["frac"])
print(key, "hbm bytes/launch %.4g" % t[key]["hbm_bytes_per_launch"])
