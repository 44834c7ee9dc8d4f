#!/usr/bin/env python3
"""Development aid: per-loop instruction mix of a gfx950 kernel from hipcc's -S output.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -x hip -Iinclude -Ilambda_amd/csrc --cuda-device-only -S -o k.s lambda_amd/csrc/lx_sweep_mq.hip
    python tools/dev/asm_loops.py k.s sweep_mq_kernelILi19ELb1

Finds the backward branches of the named kernel and prints, for every innermost loop (label .. branch back), how many
instructions of each class it holds -- enough to see whether spills (scratch_*) sit inside the steady-state loops.
"""
import re
import sys
from collections import Counter


def classify(op):
    if op.startswith("scratch_"):
        return "scratch"
    if op.startswith("v_pk_"):
        return "v_pk"
    if op.startswith("v_perm"):
        return "v_perm"
    if op.startswith("ds_"):
        return "ds"
    if op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_"):
        return "vmem"
    if "dpp" in op:
        return "dpp"
    if op.startswith("v_"):
        return "v_other"
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, key = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*%s\S*:" % re.escape(key), l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    body = lines[start:end]
    labels = {}
    instrs = []  # (index in body, op, full)
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = len(instrs)
            continue
        s = l.strip()
        if not s or s.startswith(";") or s.startswith("."):
            continue
        instrs.append((i, s.split()[0], s))
    loops = []
    for n, (_, op, full) in enumerate(instrs):
        if op.startswith("s_cbranch") or op == "s_branch":
            tgt = full.split()[-1]
            if tgt in labels and labels[tgt] <= n:
                loops.append((labels[tgt], n, tgt))
    print("%d instructions, %d backward branches" % (len(instrs), len(loops)))
    for lo, hi, tgt in sorted(loops):
        inner = not any(l2 > lo and h2 < hi for l2, h2, _ in loops) and not any((l2, h2) != (lo, hi) and l2 >= lo and h2 <= hi for l2, h2, _ in loops)
        c = Counter(classify(op) for _, op, _ in instrs[lo:hi + 1])
        dpp = sum(1 for _, _, f in instrs[lo:hi + 1] if "row_shr" in f or "wave_shr" in f or "row_bcast" in f)
        print("%-12s %5d instr %s  %s dpp=%d" % (tgt, hi - lo + 1, "inner" if inner else "outer", dict(sorted(c.items())), dpp))


if __name__ == "__main__":
    main()
