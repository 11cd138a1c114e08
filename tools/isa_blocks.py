#!/usr/bin/env python3
"""Per-basic-block instruction census of one kernel in a hipcc -save-temps .s file.

usage: isa_blocks.py file.s kernel_substring [--min-mfma N] [--dump LABEL]

Prints, for every basic block (label to label) of the first kernel whose mangled name contains the substring, the
number of MFMA / VALU / transcendental / SALU / LDS / VMEM / wait instructions, so the hot loop (the block(s) with
the MFMAs and a backward branch) can be read off and its VALU-per-MFMA ratio checked after every edit.
Not part of the product; nothing imports it.
"""
import re
import sys
from collections import Counter, OrderedDict


def classify(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith(("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt", "v_sin", "v_cos")):
        return "trans"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith("s_nop"):
        return "nop"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")):
        return "vmem"
    return "other"


def main():
    path, key = sys.argv[1], sys.argv[2]
    dump = None
    min_mfma = 0
    args = sys.argv[3:]
    while args:
        a = args.pop(0)
        if a == "--dump":
            dump = args.pop(0)
        elif a == "--min-mfma":
            min_mfma = int(args.pop(0))
    lines = open(path).read().split("\n")
    start = None
    for i, ln in enumerate(lines):
        if re.match(r"^_Z\w*:\s*(;.*)?$", ln) and key in ln:
            start = i
            break
    if start is None:
        sys.exit("kernel not found")
    blocks = OrderedDict()
    cur = "entry"
    blocks[cur] = []
    for ln in lines[start + 1:]:
        s = ln.strip()
        if s.startswith(".Lfunc_end") or s.startswith("s_endpgm"):
            if s.startswith("s_endpgm"):
                blocks[cur].append(s)
            if s.startswith(".Lfunc_end"):
                break
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", s)
        if m:
            cur = m.group(1)
            blocks[cur] = []
            continue
        if not s or s.startswith((";", ".", "//")):
            continue
        blocks[cur].append(s)
    names = list(blocks)
    order = {n: i for i, n in enumerate(names)}
    print("%-12s %5s %5s %5s %5s %5s %5s %5s %5s  %s" % ("block", "mfma", "valu", "trans", "salu", "lds", "vmem", "wait", "nop", "branches"))
    tot = Counter()
    for n, ins in blocks.items():
        c = Counter(classify(x.split()[0]) for x in ins)
        tot.update(c)
        br = []
        for x in ins:
            if x.startswith(("s_cbranch", "s_branch")):
                tgt = x.split()[-1]
                br.append(tgt + ("^" if order.get(tgt, 1 << 30) <= order[n] else ""))
        if c["mfma"] >= min_mfma:
            print("%-12s %5d %5d %5d %5d %5d %5d %5d %5d  %s" % (n, c["mfma"], c["valu"], c["trans"], c["salu"], c["lds"], c["vmem"], c["wait"], c["nop"], " ".join(br)))
    print("total", dict(tot))
    if dump:
        print("---- %s ----" % dump)
        for x in blocks[dump]:
            print("   ", x)


if __name__ == "__main__":
    main()
