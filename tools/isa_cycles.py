#!/usr/bin/env python3
"""Static issue-cycle model of a kernel's hot loop from its ISA, with the per-class costs measured by
tools/ubench/valu_clock (cycles a wave64 instruction holds its SIMD at full occupancy):
  full-rate (f32 add/sub/mul/fma, mov, and/or/xor/not, 32-bit integer add/sub)  ~2
  half-rate (compare, cndmask, min/max/med3, bfi/and_or/or3, shifts, cvt, mul_lo/hi, every FP64 add/mul/fma, packed f32)  4
  f32 transcendentals (rcp, rsq, sqrt, ...)  8;   f64 transcendentals  16
Usage: isa_cycles.py file.s kernel_symbol_substring first_line last_line   (line numbers inside the kernel, 1-based)
Prints cycles by class and the most expensive mnemonics."""
import collections
import re
import sys

FULL = ("v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_fmaak_f32", "v_fmamk_f32", "v_mov_b32",
        "v_xor_b32", "v_and_b32", "v_or_b32", "v_not_b32", "v_bitop3_b32", "v_lshrrev_b32", "v_ashrrev_i32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_accvgpr", "v_mov_b64")
T32 = ("v_rcp_f32", "v_sqrt_f32", "v_rsq_f32", "v_exp_f32", "v_log_f32", "v_sin_f32", "v_cos_f32")
T64 = ("v_rcp_f64", "v_rsq_f64", "v_sqrt_f64")


def klass(m):
    if m.startswith(T64):
        return "trans64", 16
    if m.startswith(T32):
        return "trans32", 8
    if m.startswith(FULL):
        return "full", 2
    return "half", 4


def main():
    path, sym = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and sym in l and l.split(";")[0].rstrip().endswith(":"))
    lo = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    hi = int(sys.argv[4]) if len(sys.argv) > 4 else 10 ** 9
    cyc = collections.Counter()
    cnt = collections.Counter()
    per = collections.Counter()
    salu = 0
    for k, l in enumerate(lines[start:], 1):
        if l.startswith(".Lfunc_end"):
            break
        if k < lo or k > hi:
            continue
        t = l.strip().split()
        if not t or t[0].startswith((";", ".")):
            continue
        op = t[0]
        if op.startswith("v_"):
            c, w = klass(op)
            cyc[c] += w
            cnt[c] += 1
            per[op] += w
        elif op.startswith("s_"):
            salu += 1
    tot = sum(cyc.values())
    print(f"lines {lo}..{hi}: {sum(cnt.values())} VALU, {tot} issue cycles, {salu} SALU")
    for c in ("full", "half", "trans32", "trans64"):
        print(f"  {c:8s} {cnt[c]:5d} instr {cyc[c]:6d} cycles ({100.0 * cyc[c] / max(tot, 1):.1f} %)")
    for op, w in per.most_common(28):
        print(f"    {op:24s} {w:5d}")


main()
