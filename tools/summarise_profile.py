#!/usr/bin/env python3
"""Summarise gpurun_out/prof_<tag> (written by tools/profile.sh) into profiles/<round>/<tag>.md
+ .json: kernel-trace stats, PMC counters per launch of pt::pixel_kernel, HBM traffic with the
gfx950 corrections of MI355X_MICROARCH.md (FETCH_SIZE x2 for wide coalesced reads is NOT applied
here because this kernel's reads are 40-byte scene records and 24-byte state words, not
16-B/lane streams -- the raw and the doubled value are both reported)."""
import csv
import glob
import json
import os
import sys

tag, rnd = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "r02")
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_{tag}")
dst = os.path.join(root, "profiles", rnd)
os.makedirs(dst, exist_ok=True)
out = {"tag": tag}

def newest(paths):
    """gpurun MERGES a run's files into the local gpurun_out/ without removing what an earlier run left there (rocprofv3 names
    its files by pid): only the newest file of a pass belongs to the run being summarised."""
    paths = sorted(paths, key=os.path.getmtime)
    return paths[-1:]


st = newest(glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True))
if st:
    rows = list(csv.DictReader(open(st[0])))
    out["kernel_stats"] = rows
    open(os.path.join(dst, f"{tag}_kernel_stats.csv"), "w").write(open(st[0]).read())

def pmc(sub):
    acc = {}
    for f in newest(glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True)):
        for r in csv.DictReader(open(f)):
            if "pixel_kernel" not in r["Kernel_Name"]:
                continue
            acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    # median over the launches: the SQ counters are identical from launch to launch, but FETCH_SIZE / WRITE_SIZE are chip-wide
    # TCC counters and a launch that overlaps any other traffic reads high (seen: 12.7, 79.8, 12.7, 12.7 MB)
    def med(v):
        v = sorted(v)
        return v[len(v) // 2] if len(v) % 2 else 0.5 * (v[len(v) // 2 - 1] + v[len(v) // 2])
    return {k: med(v) for k, v in acc.items()}, {k: v for k, v in acc.items() if k in ("FETCH_SIZE", "WRITE_SIZE")}

counters = {}
for sub in ("pmc_sq", "pmc_sq2", "pmc_sq3", "pmc_mix1", "pmc_mix2", "pmc_write", "pmc_fetch"):
    c, n = pmc(sub)
    counters.update(c)
    if n:
        out.setdefault("tcc_per_launch_kb", {}).update(n)
out["pmc_per_launch_avg"] = counters  # (median over launches; key name kept for the tools that read it)
if "WRITE_SIZE" in counters or "FETCH_SIZE" in counters:
    w = counters.get("WRITE_SIZE", 0.0) * 1024
    f = counters.get("FETCH_SIZE", 0.0) * 1024
    out["hbm_bytes_per_launch"] = {"write": w, "fetch_raw": f, "fetch_x2": 2 * f, "total_raw": w + f}
if counters.get("SQ_BUSY_CYCLES") and counters.get("SQ_ACTIVE_INST_VALU"):
    out["derived"] = {
        "valu_inst_per_wave": counters.get("SQ_INSTS_VALU", 0) / max(counters.get("SQ_WAVES", 1), 1),
        "active_inst_valu_over_wave_cycles": counters["SQ_ACTIVE_INST_VALU"] / max(counters.get("SQ_WAVE_CYCLES", 1), 1),
        "wait_inst_any_over_wave_cycles": counters.get("SQ_WAIT_INST_ANY", 0) / max(counters.get("SQ_WAVE_CYCLES", 1), 1),
        "wait_any_over_wave_cycles": counters.get("SQ_WAIT_ANY", 0) / max(counters.get("SQ_WAVE_CYCLES", 1), 1),
    }
# VALU issue roofline: wave-instructions per second against the guide's peak (MI355X_MICROARCH.md: a wave64 VALU
# instruction holds its SIMD-32 for 2 cycles -> 1024 SIMDs x 2.4 GHz / 2).  No per-class weights: a model built on
# them once produced a "utilisation" above 1; what is reported is count / time / peak, nothing else.
VALU_PEAK = 256 * 4 * 2.4e9 / 2.0
MIX = ["SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_TRANS_F32", "SQ_INSTS_VALU_ADD_F64",
       "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64", "SQ_INSTS_VALU_CVT", "SQ_INSTS_VALU_INT32",
       "SQ_INSTS_VALU_INT64"]
if "SQ_INSTS_VALU" in counters and st:
    kern = next(r for r in out["kernel_stats"] if "pixel_kernel" in r["Name"])
    t_ns = float(kern["AverageNs"])
    out["valu"] = {
        "insts_per_launch": counters["SQ_INSTS_VALU"], "kernel_ms": t_ns / 1e6,
        "achieved_ginst_per_s": counters["SQ_INSTS_VALU"] / t_ns, "peak_ginst_per_s": VALU_PEAK / 1e9,
        "frac": counters["SQ_INSTS_VALU"] / (t_ns * 1e-9) / VALU_PEAK,
        "flops_fp32": counters.get("SQ_INSTS_VALU_FLOPS_FP32"), "flops_fp64": counters.get("SQ_INSTS_VALU_FLOPS_FP64"),
        "mix": {k.replace("SQ_INSTS_VALU_", ""): counters[k] for k in MIX if k in counters},
    }
for log in ("stats.log",):
    p = os.path.join(src, log)
    if os.path.exists(p):
        for line in open(p):
            if line.startswith('{"metric"'):
                out["bench_line_under_profiler"] = json.loads(line)
                ki = out["bench_line_under_profiler"].get("kernel_info", {})
                out["fingerprint"] = ki.get("fingerprint")  # the build these counters belong to
                out["num_vgprs"] = ki.get("num_vgprs")
            elif line.startswith("{'spp'"):  # tools/cfg4_run.py's record
                import ast
                rec = ast.literal_eval(line.strip())
                out["fingerprint"], out["num_vgprs"], out["samples_per_launch"] = rec.get("fingerprint"), rec.get("num_vgprs"), rec.get("samples_per_launch")
json.dump(out, open(os.path.join(dst, f"{tag}.json"), "w"), indent=1)
print(json.dumps({k: out[k] for k in out if k != "bench_line_under_profiler"}, indent=1))
