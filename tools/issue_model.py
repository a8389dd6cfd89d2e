#!/usr/bin/env python3
"""Class-weighted VALU issue model of the headline kernel: how many SIMD cycles its instruction mix NEEDS at the issue
rate of each instruction class, against the SIMD cycles the kernel TOOK (GRBM_GUI_ACTIVE of the counter run).

  full-rate  (v_add/sub/mul/fma_f32, v_mov, and/or/xor/not, 32-bit integer add/sub)                       2 cycles
  half-rate  (compare, cndmask, min/max/med3, bfi/and_or/or3, shifts, cvt, every FP64 add/mul/fma, ...)         4
  f32 transcendental (rcp, rsq, sqrt)                                                                       8
  f64 transcendental                                                                                       16

2 cycles per wave64 instruction is the guide's SIMD-32 figure; the 1 : 2 : 4 : 8 ladder is what tools/ubench/valu_clock
measured on this chip (2.2 : 4.0-4.2 : 8.1 : 16.2 cycles including its loop overhead; profiles/r02/valu_issue_cost_ubench.txt).
With the architectural costs `needed` is a lower bound and frac = needed / taken cannot exceed 1.  `needed_at_measured_costs`
is the same sum with the microbenchmark's own figures, in cycles, for reference: the kernel has been seen to issue up to 2 %
FASTER than those (the microbenchmark's loop overhead is inside them), which is why they are not used for a fraction.

Counts per class come from the profiler's own counters (SQ_INSTS_VALU_{ADD,MUL,FMA}_F32 full-rate; *_F64 and CVT half-rate;
TRANS_F32 / TRANS_F64).  The counters do not split the rest (integer ops, compares, selects, min/max, moves, bit ops) into
full- and half-rate; that split is taken from the ISA of the kernel's hot path (the sample loop up to the literal-loop
fallbacks, recognised by their FP64 division), where every instruction is classified by mnemonic.
Usage: issue_model.py <isa.s> <kernel symbol substring> <profile json (profiles/rNN/tag.json)> <key in valu_roofline.json>"""
import collections, json, os, sys

isa, sym, prof, key = sys.argv[1:5]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COST = {"full": 2.0, "half": 4.0, "trans32": 8.0, "trans64": 16.0}
MEASURED = {"full": 2.2, "half": 4.1, "trans32": 8.1, "trans64": 16.2}
F32 = ("v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_fmaak_f32", "v_fmamk_f32")
FULL_OTHER = ("v_mov_b32", "v_xor_b32", "v_and_b32", "v_or_b32", "v_not_b32", "v_bitop3_b32", "v_lshrrev_b32", "v_ashrrev_i32", "v_add_u32", "v_sub_u32", "v_subrev_u32")
T32 = ("v_rcp_f32", "v_sqrt_f32", "v_rsq_f32", "v_sin_f32", "v_cos_f32", "v_exp_f32", "v_log_f32")
T64 = ("v_rcp_f64", "v_rsq_f64", "v_sqrt_f64")
F64CVT = ("v_add_f64", "v_mul_f64", "v_fma_f64", "v_fmac_f64", "v_cvt_")
lines = open(isa).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and sym in l and l.split(";")[0].rstrip().endswith(":"))
body = []
for l in lines[start + 1:]:
    if l.startswith(".Lfunc_end"):
        break
    body.append(l)
loop = next(i for i, l in enumerate(body) if "This Loop Header: Depth=1" in l)      # the sample loop
cold = next(i for i, l in enumerate(body) if i > loop and "v_div_scale_f64" in l)   # first literal-loop fallback (FP64 division)
while not body[cold].startswith(".LBB"):
    cold -= 1
rest = collections.Counter()
for l in body[loop:cold]:
    t = l.strip().split()
    if not t or not t[0].startswith("v_"):
        continue
    op = t[0]
    if op.startswith(F32 + T32 + T64 + F64CVT):
        continue
    rest["full" if op.startswith(FULL_OTHER) else "half"] += 1
f_full = rest["full"] / max(rest["full"] + rest["half"], 1)
d = json.load(open(os.path.join(root, prof)))
c, v = d["pmc_per_launch_avg"], d["valu"]
samples = d.get("samples_per_launch") or 2 ** 30
per = lambda name: c[name] / (samples / 64)
f32 = per("SQ_INSTS_VALU_ADD_F32") + per("SQ_INSTS_VALU_MUL_F32") + per("SQ_INSTS_VALU_FMA_F32")
f64cvt = per("SQ_INSTS_VALU_ADD_F64") + per("SQ_INSTS_VALU_MUL_F64") + per("SQ_INSTS_VALU_FMA_F64") + per("SQ_INSTS_VALU_CVT")
t32, t64 = per("SQ_INSTS_VALU_TRANS_F32"), per("SQ_INSTS_VALU_TRANS_F64")
total = per("SQ_INSTS_VALU")
other = total - f32 - f64cvt - t32 - t64
mix = {"full": f32 + f_full * other, "half": f64cvt + (1.0 - f_full) * other, "trans32": t32, "trans64": t64}
need = sum(mix[k] * COST[k] for k in mix)
clock = c["GRBM_GUI_ACTIVE"] / 8 / (v["kernel_ms"] * 1e-3)  # the counter sums the 8 XCDs
took = v["kernel_ms"] * 1e-3 * clock * 1024 / (samples / 64)
rec = {"costs_cycles": COST, "mix_per_sample": {k: round(x, 1) for k, x in mix.items()}, "valu_per_sample": round(total, 1),
       "unclassified_by_pmc_per_sample": round(other, 1), "full_rate_share_of_unclassified_from_isa": round(f_full, 3),
       "needed_simd_cycles_per_sample": round(need, 1), "clock_ghz": round(clock / 1e9, 4),
       "taken_simd_cycles_per_sample": round(took, 1), "frac": round(need / took, 4) if need <= took else None,
       "needed_at_measured_costs": round(sum(mix[k] * MEASURED[k] for k in mix), 1), "measured_costs_cycles": MEASURED,
       "method": "tools/issue_model.py: architectural class costs (2-cycle SIMD-32 issue, 1:2:4:8 ladder measured by tools/ubench/valu_clock)"}
vp = os.path.join(root, "profiles", "valu_roofline.json")
r = json.load(open(vp))
r[key]["class_weighted"] = rec
json.dump(r, open(vp, "w"), indent=1)
print(json.dumps(rec))
