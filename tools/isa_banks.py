#!/usr/bin/env python3
"""VGPR operand-bank conflicts in a kernel's hot path.  Measured on gfx950 (tools/ubench/valu_banks.hip, valu_banks_sweep):
a VALU instruction that reads THREE VGPRs issues at half rate (4.1 instead of 2.2 cycles) exactly when all three register
numbers have the same parity (v_fmac/v_fmamk-style forms: the destination is the third read); two-source instructions and
inline constants never conflict.  Usage: isa_banks.py <isa.s> <kernel symbol substring>"""
import collections, re, sys
isa, sym = sys.argv[1:3]
lines = open(isa).read().split("\n")
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
three = collections.Counter(); conf = collections.Counter()
ACC = ("v_fmac_f32", "v_fmac_f64", "v_fmac_f16")  # destination is read
for l in body[loop:cold]:
    t = l.split(";")[0].strip()
    if not t.startswith("v_"):
        continue
    op = t.split()[0].replace("_e32", "").replace("_e64", "")
    args = t[len(t.split()[0]):]
    ops = [a.strip() for a in args.split(",")]
    if not ops:
        continue
    dst, srcs = ops[0], ops[1:]
    regs = []
    for a in srcs:
        m = re.search(r"\bv(\d+)\b", a)            # single VGPR (with |..| or - modifiers)
        m2 = re.search(r"v\[(\d+):(\d+)\]", a)      # 64-bit pair: use the low register
        if m2: regs.append(int(m2.group(1)))
        elif m: regs.append(int(m.group(1)))
    if op in ACC:
        m = re.search(r"\bv(\d+)\b", dst) or re.search(r"v\[(\d+):", dst)
        if m: regs.append(int(m.group(1)))
    if "cndmask" in op:
        continue
    if len(regs) >= 3:
        three[op] += 1
        if len({r & 1 for r in regs[:3]}) == 1:
            conf[op] += 1
print("three-VGPR-source instructions:", sum(three.values()), " all of one parity:", sum(conf.values()))
for k, v in three.most_common():
    print(f"  {k:18s} {v:4d}  conflicts {conf[k]:4d}")
