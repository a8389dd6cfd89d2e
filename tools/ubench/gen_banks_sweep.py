#!/usr/bin/env python3
"""Writes valu_banks_sweep.hip from valu_banks.hip: v_fma_f32 with src0 = v28 and every (src1, src2) in v29..v36 x v29..v39, plus a
few odd src0 -- the sweep behind profiles/r03/valu_banks_sweep.txt (slow exactly when all three source VGPRs are even-numbered;
by symmetry and the all-odd rows of valu_banks.hip: when all three have the same parity).
  python3 gen_banks_sweep.py && hipcc --offload-arch=gfx950 -O2 -o valu_banks_sweep valu_banks_sweep.hip && ./valu_banks_sweep 32768"""
import os, re
here = os.path.dirname(os.path.abspath(__file__))
s = open(os.path.join(here, "valu_banks.hip")).read()
head = s[:s.index("KERNEL(add_2banks")]
ks, es = [], []
for s1 in range(29, 37):
    for s2 in range(29, 40):
        if s2 != s1:
            ks.append(f'KERNEL(f_{s1}_{s2}, "v_fma_f32", "v28, v{s1}, v{s2}")'); es.append(f"E(f_{s1}_{s2})")
for s0 in (29, 30, 31):
    for s1 in (32, 33, 34, 35):
        ks.append(f'KERNEL(g_{s0}_{s1}_36, "v_fma_f32", "v{s0}, v{s1}, v36")'); es.append(f"E(g_{s0}_{s1}_36)")
main = re.sub(r"Entry es\[\] = \{.*?\};", "Entry es[] = {" + ", ".join(es) + "};", s[s.index("typedef void (*kfn)"):], flags=re.S)
open(os.path.join(here, "valu_banks_sweep.hip"), "w").write(head + "\n".join(ks) + "\n" + main)
