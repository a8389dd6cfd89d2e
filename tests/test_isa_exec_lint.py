"""tools/isa_exec_lint.py, the check the library's Makefile runs over the assembly of every build (EXACTNESS.md A.12): vector copies
between the head of a basic block and the `s_or_b64 exec, exec, ...` that re-opens the lanes which sat out the region before it.
The lint itself on the two shapes (the miscompiled join as it was found in variant 13's philox build, and a well-formed join),
and over the assembly the last `make` left under cuda-pathtrace_amd/csrc/build/ -- the exact instruction stream of the
libraries the other tests load."""
import glob
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("isa_exec_lint", os.path.join(ROOT, "tools", "isa_exec_lint.py"))
lint_mod = importlib.util.module_from_spec(spec)
spec.loader.exec_module(lint_mod)

BAD = """
_ZN2pt12pixel_kernelILi1ELi13ELb1ELi0EEEv15PixelKernelArgs: ; @kernel
	s_and_saveexec_b64 s[16:17], s[0:1]
	s_cbranch_execnz .LBB20_270
.LBB20_138:                             ;   in Loop: Header=BB20_96 Depth=1
	v_mov_b32_e32 v35, v52
	v_mov_b32_e32 v50, v113
	s_mov_b32 s89, s39
	s_or_b64 exec, exec, s[16:17]
	s_branch .LBB20_140
.Lfunc_end20:
"""
GOOD = """
_ZN2pt12pixel_kernelILi0ELi13ELb1ELi0EEEv15PixelKernelArgs: ; @kernel
	s_and_saveexec_b64 s[16:17], s[0:1]
	s_cbranch_execz .LBB21_138
; %bb.137:
	v_mul_f32_e32 v1, v2, v3
.LBB21_138:                             ;   in Loop: Header=BB21_96 Depth=1
	s_or_b64 exec, exec, s[16:17]
	v_mov_b32_e32 v35, v52
	v_mov_b32_e32 v50, v113
	s_branch .LBB21_140
.LBB21_270:                             ; an out-of-line block that ends its own region: computation, not copies
	v_mul_f32_e32 v18, 0x4f800000, v86
	v_mov_b32_e32 v19, v18
	s_or_b64 exec, exec, s[20:21]
	s_branch .LBB21_138
.Lfunc_end21:
"""


def test_lint_flags_copies_in_front_of_the_exec_restore(tmp_path):
    bad, good = tmp_path / "bad.s", tmp_path / "good.s"
    bad.write_text(BAD)
    good.write_text(GOOD)
    found = lint_mod.lint(str(bad))
    assert len(found) == 1 and found[0][1] == ".LBB20_138" and len(found[0][3]) == 2
    assert lint_mod.lint(str(good)) == []


def test_built_libraries_are_clean():
    files = sorted(glob.glob(os.path.join(ROOT, "cuda-pathtrace_amd", "csrc", "build", "*", "*-gfx950.s")))
    if not files:
        pytest.skip("no build directory here (the libraries were built elsewhere): the Makefile ran the lint when they were")
    flavours = {os.path.basename(os.path.dirname(f)) for f in files}
    assert {"prod", "lab"} <= flavours, flavours
    kernels = 0
    for f in files:
        assert lint_mod.lint(f) == [], f
        kernels += sum(1 for line in open(f) if line.startswith("_Z") and "pixel_kernel" in line and line.split(";")[0].rstrip().endswith(":"))
    assert kernels > 20  # (the assembly really is the kernels': both libraries' pixel_kernel instantiations)


def test_pooled_push_is_three_separate_descending_stores():
    """The push of variant 13's pooled walk (csrc/pt_grid.h, grid_trips_pooled (1); EXACTNESS.md A.9 (vi)) relies on its three ring
    stores going out as three separate ds_write_b32 in DESCENDING slot order: a lane with fewer than three spheres writes garbage
    into slots a later store of the same push overwrites -- a ds_write2_b32 / ds_write_b64 fusing two of them would let a lane's
    garbage race its neighbour's rightful value inside one instruction (nothing pinned this).  Read from the assembly
    of the installed builds: every block that computes ring positions (six chained v_mbcnt) and stores must show exactly that."""
    import re

    files = sorted(glob.glob(os.path.join(ROOT, "cuda-pathtrace_amd", "csrc", "build", "*", "pt_kernel-*-gfx950.s")))
    if not files:
        pytest.skip("no build directory here (the libraries were built elsewhere)")
    pushes = 0
    for f in files:
        kernel, block = None, []

        def check(block):
            n = 0
            mb = sum(1 for i in block if i.startswith("v_mbcnt"))
            st = [i for i in block if i.startswith("ds_write")]
            if mb >= 6 and st:
                assert all(i.split()[0] == "ds_write_b32" for i in st), (f, kernel, st)
                assert len(st) == 3, (f, kernel, st)
                regs = {i.split()[1].rstrip(",") for i in st}
                offs = [int(re.search(r"offset:(\d+)", i).group(1)) if "offset:" in i else 0 for i in st]
                assert len(regs) == 1 and offs[0] == offs[1] + 4 == offs[2] + 8, (f, kernel, st)
                n = 1
            return n

        for line in open(f):
            s = line.strip()
            m = re.match(r"(_Z\w+):", s)
            if m:
                kernel, block = (m.group(1) if "Li13E" in m.group(1) else None), []
                continue
            if kernel is None:
                continue
            if s.startswith(".Lfunc_end"):
                pushes += check(block)
                kernel = None
            elif re.match(r"\.LBB\d+_\d+:", s):
                pushes += check(block)
                block = []
            elif s and not s.startswith((";", ".")):
                block.append(s.split(";")[0].strip())
    assert pushes >= 2 * 2 * 3  # two libraries x two generators x (two pushes of a lock-step round + the sweep's)
