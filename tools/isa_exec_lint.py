#!/usr/bin/env python3
"""Finds vector instructions placed between the start of a basic block and the `s_or_b64 exec, exec, s[..]` that re-opens the
lanes which sat out the preceding divergent region.  The register allocator's live-range splitting of this compiler has been
seen to put the copies of a split there (EXACTNESS.md A.12): they then run for the lanes that took the region only -- for NO lane
when the region was skipped -- and the value the other lanes carry is lost.  Usage: isa_exec_lint.py file.s...
(the library's Makefile runs it over the assembly of every build; exit status 1 = found, 2 = nothing to read;
PT_SKIP_ISA_LINT=1 in the environment skips it.  A HEURISTIC: it knows the one shape seen so far -- a block head that holds nothing
but register copies / spill traffic before the restore -- and the parity suite and soaks remain the gate for everything else.)"""
import re, sys

def lint(path, only=None):
    found = []
    kernel = None
    pending = None  # vector writes seen since the last label, before anything that ends the block's head
    label = None
    for ln, line in enumerate(open(path), 1):
        s = line.strip()
        if not s or s.startswith(";") or s.startswith("."):
            if re.match(r"\.LBB\d+_\d+:", s):
                label, pending = s.split(":")[0], []
            continue
        m = re.match(r"(_Z\w+):", s)
        if m:
            kernel, pending, label = m.group(1), None, None
            continue
        if re.match(r"\.?\w+:", s):
            continue
        if pending is None:
            continue
        op = s.split()[0]
        if op == "s_or_b64" and re.match(r"s_or_b64\s+exec,\s*exec,", s):
            # copies and spill traffic only: computation there belongs to the region that is ending (an out-of-line block
            # that closes its own region), which is how the compiler normally lays such blocks out
            # (any v_mov -- from a vector or scalar register or a constant: a rematerialised value is a split's copy too)
            copies = pending and all(re.match(r"(v_mov_b32_e32 v\d+, |v_mov_b64_e32 v\[[\d:]+\], |v_accvgpr_\w+ |scratch_(load|store)_)", i) for i in pending)
            if copies and (only is None or (kernel and only in kernel)):
                found.append((kernel, label, ln, pending[:]))
            pending = None
        elif op.startswith(("v_", "ds_", "global_", "scratch_", "buffer_", "flat_")):
            if not op.startswith(("v_readlane", "v_readfirstlane", "v_cmp")):
                pending.append(s.split(";")[0].strip())
        elif op.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_and_saveexec", "s_or_saveexec", "s_andn2_b64", "s_xor_b64", "s_setpc")):
            pending = None
    return found

if __name__ == "__main__":
    import glob, os
    if os.environ.get("PT_SKIP_ISA_LINT") == "1":  # escape hatch: a heuristic on text must never make the library unbuildable
        print("isa_exec_lint: skipped (PT_SKIP_ISA_LINT=1); the parity tests and soaks are the gate")
        sys.exit(0)
    paths = [p for a in sys.argv[1:] for p in (glob.glob(a) or [a])]
    missing = [p for p in paths if not os.path.exists(p)]
    if missing or not paths:
        print(f"isa_exec_lint: no assembly to read ({missing or 'no arguments'}): was the library compiled with -save-temps for this "
              f"--offload-arch?  Set PT_SKIP_ISA_LINT=1 to build without the lint.")
        sys.exit(2)
    total = 0
    for path in paths:
        res = lint(path)
        total += len(res)
        for k, lab, ln, ins in res:
            print(f"{path}: {k} {lab} (line {ln}): {len(ins)} copy/spill instruction(s) before the exec restore: {ins[:4]}")
    print(f"isa_exec_lint: {len(paths)} file(s), {total} suspect block head(s)")
    sys.exit(1 if total else 0)
