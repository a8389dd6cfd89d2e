#!/usr/bin/env python3
"""oneminus_f32_nb (pt_device.h) against (float)sqrt(1.0 - (double)(ry*ry)) for every float ry in [0, 1], and how many
of them it flags for the literal redo (lab library)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
lab = ge.load_lab(); lab.set_device(0)
n = 0x3F800001
bad, ex = lab.unary_compare(lab.FN_ONEMINUS_F32, lab.FN_ONEMINUS_LITERAL, 0, n)
flag, exf = lab.unary_compare(lab.FN_ONEMINUS_F32_FLAG, lab.FN_ZERO, 0, n)
print(f"{n} inputs: {bad} differ (e.g. 0x{ex:08x}), {flag} flagged = {flag / n:.3e} (e.g. 0x{exf:08x})")
