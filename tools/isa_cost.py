#!/usr/bin/env python3
"""Static cost breakdown of a gfx950 kernel's ISA by basic block, weighted with the issue
costs measured by tools/ubench/valu_cost (units of one v_add_f32).  Usage:
  isa_cost.py file.s kernel_symbol_substring"""
import re
import sys

COST3 = ("v_rcp_f32", "v_sqrt_f32", "v_rsq_f32", "v_exp_f32", "v_log_f32", "v_sin_f32", "v_cos_f32")
COST6 = ("v_rcp_f64", "v_rsq_f64", "v_sqrt_f64")
FAST = ("v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_mov_b32", "v_xor_b32", "v_add_u32", "v_sub_u32",
        "v_subrev_u32", "v_and_b32", "v_or_b32", "v_fmaak_f32", "v_fmamk_f32", "v_accvgpr", "v_not_b32", "v_bitop3_b32", "v_lshrrev_b32", "v_ashrrev_i32", "v_mov_b64")


def cost(m):
    if m.startswith(COST6):
        return 6.0
    if m.startswith(COST3):
        return 3.0
    if m.startswith(FAST):
        return 1.0
    return 1.55


def main():
    path, sym = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and sym in l and l.split("#")[0].split(";")[0].rstrip().endswith(":"))
    blocks, cur = [], None
    for l in lines[start + 1:]:
        if l.startswith(".Lfunc_end"):
            break
        m = re.match(r"^(\.LBB\S+|; %bb\.\d+):?", l.strip())
        if m:
            cur = {"name": m.group(1), "depth": 0, "valu": 0, "cost": 0.0, "salu": 0, "mem": 0, "mn": {}}
            blocks.append(cur)
            d = re.search(r"Depth=(\d+)", l)
            if d:
                cur["depth"] = int(d.group(1))
            continue
        if cur is None:
            cur = {"name": "entry", "depth": 0, "valu": 0, "cost": 0.0, "salu": 0, "mem": 0, "mn": {}}
            blocks.append(cur)
        d = re.search(r"Depth=(\d+)", l)
        if d and l.strip().startswith(";"):
            cur["depth"] = max(cur["depth"], int(d.group(1)))
            continue
        t = l.strip().split()
        if not t or t[0].startswith(";") or t[0].startswith("."):
            continue
        op = t[0]
        if op.startswith("v_"):
            cur["valu"] += 1
            cur["cost"] += cost(op)
            cur["mn"][op] = cur["mn"].get(op, 0) + 1
        elif op.startswith("s_"):
            cur["salu"] += 1
        elif op.startswith(("ds_", "global_", "buffer_", "flat_", "scratch_")):
            cur["mem"] += 1
    tot = {}
    for b in blocks:
        k = b["depth"]
        tot.setdefault(k, [0, 0.0, 0])
        tot[k][0] += b["valu"]
        tot[k][1] += b["cost"]
        tot[k][2] += b["salu"]
    verbose = len(sys.argv) > 3
    for b in blocks:
        if verbose and b["valu"]:
            top = sorted(b["mn"].items(), key=lambda kv: -kv[1])[:6]
            print(f"{b['name']:12s} depth {b['depth']} valu {b['valu']:4d} cost {b['cost']:7.1f} salu {b['salu']:3d} mem {b['mem']:2d}  {top}")
    print("by loop depth (static): depth -> VALU count, weighted cost, SALU")
    for k in sorted(tot):
        print(f"  depth {k}: {tot[k][0]:5d} VALU  {tot[k][1]:8.1f} units  {tot[k][2]:4d} SALU")


main()
