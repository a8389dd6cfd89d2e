"""Guard for the instruction-rate work of round 3 (profiles/r03/valu_clock_bit_ops.txt, valu_banks.txt): the
headline kernel's hot path -- from the sample loop's header to the first literal fallback, the region tools/isa_lines.py prices --
must not fall back to the half-rate forms the compiler prefers.  Compiles pt_kernel.hip to gfx950 assembly on the CPU (no GPU).

Instruction counts are a property of (source, compiler), so the yardstick is a committed baseline PER COMPILER VERSION
(tests/golden/isa_baseline.json; `python tests/test_isa_rates.py --record` adds the record of the installed hipcc): the test
fails when the same compiler now makes something worse of the source, and skips -- printing what it measured -- under a compiler
that has no record yet.  A new ROCm is not a failure; it is a reason to look at the numbers and record a baseline."""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BASELINE = os.path.join(ROOT, "tests", "golden", "isa_baseline.json")
KERNEL = "pixel_kernelILi0ELi6ELb0ELi5E"  # pixel_kernel<XORWOW, 6, false, 5>: the headline build
HALF = ("v_cmp", "v_cndmask", "v_min", "v_max", "v_med3", "v_bfi", "v_and_or", "v_or3", "v_lshl", "v_add3", "v_cvt", "v_sqrt", "v_rcp", "v_rsq",
        "v_mul_f64", "v_add_f64", "v_fma_f64", "v_fmac_f64", "v_ldexp", "v_rndne", "v_div", "v_readfirstlane", "v_addc", "v_mul_lo", "v_mul_hi",
        "v_mad", "v_xad", "v_bfe")
FORMS = ("v_bfi_b32", "v_and_or_b32", "v_or3_b32", "v_med3_f32", "v_lshlrev_b32", "v_sqrt_f32", "v_rcp_f32")  # half-rate (or slower)


def compiler_id():
    out = subprocess.run(["/opt/rocm/bin/hipcc", "--version"], capture_output=True, text=True).stdout
    m = re.search(r"HIP version: (\S+)", out)
    return m.group(1) if m else "unknown"


def extract_hot_path():
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


def metrics(hot_path):
    ops = collections.Counter(t.split()[0].replace("_e32", "").replace("_e64", "") for t in hot_path)
    sgpr = 0  # full-rate instructions with an SGPR source operand (they then issue at half rate)
    for t in hot_path:
        op = t.split()[0]
        if op.replace("_e32", "").replace("_e64", "").startswith(HALF):
            continue
        srcs = [a.strip() for a in t[len(op):].split(",")][1:]
        if any(re.match(r"^[-|]*s(\d+|\[)", a) for a in srcs):
            sgpr += 1
    rec = {"valu": len(hot_path), "full_rate_with_sgpr_operand": sgpr, "v_bitop3_b32": ops["v_bitop3_b32"]}
    rec.update({k: ops[k] for k in FORMS})
    return rec


@pytest.fixture(scope="module")
def hot_path():
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc")
    return extract_hot_path()


def test_hot_path_against_the_baseline_of_this_compiler(hot_path):
    got, cid = metrics(hot_path), compiler_id()
    base = json.load(open(BASELINE)).get(cid) if os.path.exists(BASELINE) else None
    if base is None:
        pytest.skip(f"no ISA baseline for hipcc {cid}; measured {got} -- record it with: python tests/test_isa_rates.py --record")
    assert got["valu"] <= base["valu"] * 1.02, (got, base)                       # not more instructions ...
    for form in FORMS:
        assert got[form] <= base[form], (form, got, base)                        # ... nor more of the half-rate forms
    assert got["v_bitop3_b32"] >= base["v_bitop3_b32"] - 4, (got, base)          # copysign, key packing, sin/cos, xor3 stay full-rate
    # constants of the per-sphere code live in VGPRs (vgpr_const): an SGPR source makes any VALU instruction half-rate
    assert got["full_rate_with_sgpr_operand"] <= base["full_rate_with_sgpr_operand"] + 4, (got, base)


def test_round3_properties_hold_whatever_the_compiler(hot_path):
    """What the source ASKS for by name and no compiler version may undo: no v_bfi / v_and_or / v_or3 / v_med3_f32 in the hot path
    (bitop3<TT>() and the integer forms replace them), and the three-operand bit function is there."""
    got = metrics(hot_path)
    for form in ("v_bfi_b32", "v_and_or_b32", "v_or3_b32", "v_med3_f32"):
        assert got[form] == 0, (form, got)
    assert got["v_bitop3_b32"] >= 50, got


if __name__ == "__main__":
    if "--record" in sys.argv:
        rec = json.load(open(BASELINE)) if os.path.exists(BASELINE) else {}
        rec[compiler_id()] = metrics(extract_hot_path())
        json.dump(rec, open(BASELINE, "w"), indent=1, sort_keys=True)
        print(json.dumps(rec, indent=1, sort_keys=True))
