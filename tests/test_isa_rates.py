"""Guard for the instruction-rate work of round 3 (profiles/r03/valu_clock_bit_ops.txt, valu_banks.txt): the
headline kernel's hot path -- from the sample loop's header to the first literal fallback, the region tools/isa_lines.py prices --
must not fall back to the half-rate forms the compiler prefers.  Compiles pt_kernel.hip to gfx950 assembly on the CPU (no GPU)."""
import collections
import os
import re
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL = "pixel_kernelILi0ELi6ELb0ELi5E"  # pixel_kernel<XORWOW, 6, false, 5>: the headline build
HALF = ("v_cmp", "v_cndmask", "v_min", "v_max", "v_med3", "v_bfi", "v_and_or", "v_or3", "v_lshl", "v_add3", "v_cvt", "v_sqrt", "v_rcp", "v_rsq",
        "v_mul_f64", "v_add_f64", "v_fma_f64", "v_fmac_f64", "v_ldexp", "v_rndne", "v_div", "v_readfirstlane", "v_addc", "v_mul_lo", "v_mul_hi",
        "v_mad", "v_xad", "v_bfe")


@pytest.fixture(scope="module")
def hot_path():
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc")
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "pt_kernel.s")
        subprocess.run(["bash", os.path.join(ROOT, "tools", "isa.sh"), out], check=True, timeout=900, capture_output=True)
        lines = open(out).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and KERNEL in l and l.split(";")[0].rstrip().endswith(":"))
    body = []
    for l in lines[start + 1:]:
        if l.startswith(".Lfunc_end"):
            break
        body.append(l)
    loop = next(i for i, l in enumerate(body) if "This Loop Header: Depth=1" in l)
    cold = next(i for i, l in enumerate(body) if i > loop and "v_div_scale_f64" in l)
    while not body[cold].startswith(".LBB"):
        cold -= 1
    return [l.split(";")[0].strip() for l in body[loop:cold] if l.strip().startswith("v_")]


def test_hot_path_uses_the_full_rate_forms(hot_path):
    ops = collections.Counter(t.split()[0].replace("_e32", "").replace("_e64", "") for t in hot_path)
    assert 2500 < len(hot_path) < 2900, len(hot_path)             # 2 790 VALU instructions (2 937 before the bit-operation work)
    for half_rate_form in ("v_bfi_b32", "v_and_or_b32", "v_or3_b32", "v_med3_f32"):
        assert ops[half_rate_form] == 0, (half_rate_form, ops[half_rate_form])
    assert ops["v_bitop3_b32"] >= 100                             # copysign, key packing, sin/cos, xor3
    assert ops["v_lshlrev_b32"] <= 24                             # XORWOW's << 4 only: << 1 is an add
    assert ops["v_sqrt_f32"] <= 50 and ops["v_rcp_f32"] <= 55     # one of each per sphere screen


def test_hot_path_keeps_sgpr_operands_out_of_full_rate_instructions(hot_path):
    """Any VALU instruction with an SGPR source issues at half rate: constants of the per-sphere code live in VGPRs (vgpr_const)."""
    n = 0
    for t in hot_path:
        op = t.split()[0]
        base = op.replace("_e32", "").replace("_e64", "")
        if base.startswith(HALF):
            continue
        srcs = [a.strip() for a in t[len(op):].split(",")][1:]
        if any(re.match(r"^[-|]*s(\d+|\[)", a) for a in srcs):
            n += 1
    assert n <= 30, n  # 22: the primary ray's basis vectors and a few loop invariants (80 with the sign mask in an SGPR)
