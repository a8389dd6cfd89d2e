#!/usr/bin/env python3
"""Post-register-allocation renaming pass against VGPR bank conflicts (experiment of round 4; profiles/r04/regbank_ab.txt).

Measured on gfx950 (tools/ubench/valu_banks.hip): a VALU instruction that reads THREE VGPRs issues at half rate exactly when all
three register numbers have the same parity (v_fmac: the destination is the third read).  hipcc allocates without regard to
parity: 60 of the headline hot loop's 243 three-source FP32 instructions conflict (tools/isa_banks.py).

This pass applies a PERMUTATION of VGPR names to a whole kernel: every occurrence of a register is renamed consistently, so the
program is the same program -- as long as the permuted registers are never part of a multi-register operand (v[a:b]: 64-bit
values, wide loads and stores need consecutive, even-aligned registers) and are not the registers the hardware initialises
(v0-v2: work-item ids).  Those stay where they are.  Which permutation: local search over swaps of one even-numbered and one
odd-numbered free register, minimising the number of conflicting three-source instructions in the hot loop.

Usage: isa_regbank.py <in.s> <out.s> <kernel symbol substring> [...]      (prints before / after per kernel)"""
import collections, re, sys

TOK = re.compile(r"(?<![\w\[:.])v(\d+)(?![\w\]:])")  # a single VGPR operand (not inside v[a:b], not part of a name)
TUP = re.compile(r"v\[(\d+):(\d+)\]")
ACC = ("v_fmac_f32", "v_fmac_f64", "v_fmac_f16", "v_fmac_f32_e32", "v_fmac_f32_e64")


def split_inst(line):
    code = line.split(";")[0]
    t = code.strip()
    if not t or t.startswith(".") or t.endswith(":") or not re.match(r"^[a-z]", t):
        return None
    parts = t.split(None, 1)
    return parts[0], (parts[1] if len(parts) > 1 else "")


def three_source_regs(op, args):
    """VGPR numbers (as written) this instruction reads through its three-source path, or None."""
    if not op.startswith("v_") or "cndmask" in op:
        return None
    ops = [a.strip() for a in args.split(",")]
    if not ops:
        return None
    dst, srcs = ops[0], ops[1:]
    regs = []
    for a in srcs:
        m2 = TUP.search(a)
        m = TOK.search(a)
        if m2:
            regs.append(("t", int(m2.group(1))))
        elif m:
            regs.append(("s", int(m.group(1))))
    if op.replace("_e32", "").replace("_e64", "") in ACC:
        m = TOK.search(dst) or TUP.search(dst)
        if m:
            regs.append(("s" if TOK.search(dst) else "t", int(m.group(1))))
    return regs[:3] if len(regs) >= 3 else None


def process(lines, start, end, sym):
    body = lines[start:end]
    pinned = {0, 1, 2}
    used = set()
    has_call = False
    for l in body:
        it = split_inst(l)
        if not it:
            continue
        op, args = it
        if op.startswith("s_swappc") or op.startswith("s_call"):
            has_call = True
        for a, b in TUP.findall(args):
            pinned.update(range(int(a), int(b) + 1))
        used.update(int(x) for x in TOK.findall(args))
    if has_call:
        print(f"{sym}: contains calls, left alone")
        return
    free = sorted(used - pinned)
    # hot loop: from the first depth-1 loop header to the first cold block (the literal fallback: v_div_scale_f64), as
    # tools/isa_banks.py and tests/test_isa_rates.py define it
    try:
        loop = next(i for i, l in enumerate(body) if "This Loop Header: Depth=1" in l)
        cold = next(i for i, l in enumerate(body) if i > loop and "v_div_scale_f64" in l)
        while not body[cold].startswith(".LBB"):
            cold -= 1
    except StopIteration:
        loop, cold = 0, len(body)
    insts = []
    for l in body[loop:cold]:
        it = split_inst(l)
        if it:
            r = three_source_regs(*it)
            if r:
                insts.append(r)
    perm = {r: r for r in used | pinned}

    def parity(kind_reg):
        kind, r = kind_reg
        return (perm[r] if kind == "s" and r in perm else r) & 1

    def conflicts(sel=None):
        return sum(1 for r in (sel if sel is not None else insts) if len({parity(x) for x in r}) == 1)

    by_reg = collections.defaultdict(list)
    for r in insts:
        for kind, x in r:
            if kind == "s":
                by_reg[x].append(r)
    before = conflicts()
    improved = True
    while improved:
        improved = False
        for a in free:
            for b in free:
                if a >= b or (perm[a] & 1) == (perm[b] & 1):
                    continue
                touched = by_reg[a] + by_reg[b]
                if not touched:
                    continue
                c0 = conflicts(touched)
                perm[a], perm[b] = perm[b], perm[a]
                c1 = conflicts(touched)
                if c1 < c0:
                    improved = True
                else:
                    perm[a], perm[b] = perm[b], perm[a]
    after = conflicts()
    moved = sum(1 for r in free if perm[r] != r)
    print(f"{sym}: {len(insts)} three-source instructions in the hot loop, conflicts {before} -> {after}; {moved} of {len(free)} free registers renamed "
          f"({len(pinned & used)} pinned)")
    for i in range(start, end):
        l = lines[i]
        it = split_inst(l)
        if not it:
            continue
        code, sep, comment = l.partition(";")
        lines[i] = TOK.sub(lambda m: "v%d" % perm.get(int(m.group(1)), int(m.group(1))), code) + sep + comment


def main():
    src, dst, syms = sys.argv[1], sys.argv[2], sys.argv[3:]
    lines = open(src).read().split("\n")
    i = 0
    while i < len(lines):
        l = lines[i]
        if l.startswith("_Z") and l.split(";")[0].rstrip().endswith(":") and any(s in l for s in syms):
            end = next(j for j in range(i + 1, len(lines)) if lines[j].startswith(".Lfunc_end"))
            process(lines, i + 1, end, l.split(":")[0][:60])
            i = end
        i += 1
    open(dst, "w").write("\n".join(lines))


if __name__ == "__main__":
    main()
