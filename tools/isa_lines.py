#!/usr/bin/env python3
"""Static issue-cycle profile of a kernel's hot path by SOURCE LINE (ISA emitted with -gline-tables-only):
every VALU instruction of the hot path (from the sample loop's header to the first literal-loop fallback, like
tools/issue_model.py) is priced with the measured class costs and charged to the .loc it carries.
Usage: isa_lines.py <isa.s> <kernel symbol substring> [top=40]
  hipcc ... -gline-tables-only --cuda-device-only -S pt_kernel.hip -o isa.s   (flags of tools/isa.sh)"""
import collections, re, sys
isa, sym = sys.argv[1:3]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
COST = {"full": 2.2, "half": 4.0, "trans32": 8.1, "trans64": 16.2}
FULL = ("v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_fmaak_f32", "v_fmamk_f32",
        "v_mov_b32", "v_xor_b32", "v_and_b32", "v_or_b32", "v_not_b32", "v_bitop3_b32", "v_lshrrev_b32", "v_ashrrev_i32", "v_add_u32", "v_sub_u32", "v_subrev_u32")
T32 = ("v_rcp_f32", "v_sqrt_f32", "v_rsq_f32", "v_sin_f32", "v_cos_f32", "v_exp_f32", "v_log_f32")
T64 = ("v_rcp_f64", "v_rsq_f64", "v_sqrt_f64")
lines = open(isa).read().split("\n")
files = {}
for l in lines:
    m = re.match(r'\s+\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', l)
    if m:
        files[int(m.group(1))] = m.group(2)
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and sym in l and l.split(";")[0].rstrip().endswith(":"))
body = []
for l in lines[start + 1:]:
    if l.startswith(".Lfunc_end"):
        break
    body.append(l)
loop = next(i for i, l in enumerate(body) if "This Loop Header: Depth=1" in l)
cold = next(i for i, l in enumerate(body) if i > loop and "v_div_scale_f64" in l)
while not body[cold].startswith(".LBB"):
    cold -= 1
by_line, by_file, total, n = collections.Counter(), collections.Counter(), 0.0, 0
cnt = collections.Counter()
cur = ("?", 0)
for l in body[loop:cold]:
    m = re.match(r'\s+\.loc\s+(\d+)\s+(\d+)', l)
    if m:
        cur = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
        continue
    t = l.strip().split()
    if not t or not t[0].startswith("v_"):
        continue
    op = t[0]
    k = "trans64" if op.startswith(T64) else "trans32" if op.startswith(T32) else "full" if op.startswith(FULL) else "half"
    by_line[cur] += COST[k]
    by_file[cur[0]] += COST[k]
    cnt[cur] += 1
    total += COST[k]
    n += 1
print(f"hot path: {n} VALU instructions, {total:.0f} issue cycles (static: every instruction once)")
for f, c in by_file.most_common():
    print(f"  {f:24s} {c:8.0f} {100 * c / total:5.1f} %")
src = {}
for (f, ln), c in by_line.most_common(top):
    if f not in src:
        try:
            src[f] = open(f"cuda-pathtrace_amd/csrc/{f}").read().split("\n")
        except OSError:
            src[f] = []
    text = src[f][ln - 1].strip()[:110] if 0 < ln <= len(src[f]) else ""
    print(f"{c:7.0f} {100 * c / total:5.1f} % {cnt[(f, ln)]:4d}  {f}:{ln}  {text}")
