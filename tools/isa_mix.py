#!/usr/bin/env python3
"""Static instruction mix of one kernel in hipcc's assembly output (hipcc -S --cuda-device-only).
usage: isa_mix.py file.s kernel-name-substring [--loops]
Counts the instructions between the kernel's label and its s_endpgm by class; with --loops, per basic block that
ends in a backward branch (the bodies of the blind-rotation step loops)."""
import collections
import re
import sys


def classify(op):
    if op.startswith(("v_fma_f64", "v_mul_f64", "v_add_f64", "v_rndne_f64", "v_fmac_f64")):
        return "valu_f64"
    if op.startswith(("v_cvt_f64", "v_cvt_i32_f64", "v_cvt_u32_f64")):
        return "valu_cvt64"
    if op.startswith("v_mfma") or op.startswith("v_smfma"):
        return "mfma"
    if op.startswith("v_accvgpr"):
        return "valu_acc_move"
    if op.startswith("v_"):
        return "valu_other"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "s_waitcnt"
    if op.startswith("s_barrier"):
        return "s_barrier"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, name = sys.argv[1], sys.argv[2]
    loops = "--loops" in sys.argv
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if name in l and not l.startswith((".", "\t")) and l.split(";")[0].strip().endswith(":"))
    body = []
    for l in lines[start + 1:]:
        body.append(l)
        if l.strip().startswith("s_endpgm"):
            break
    label = re.compile(r"^(\.LBB[0-9_]+):")
    blocks, cur, order = collections.OrderedDict(), "entry", {}
    blocks[cur] = []
    for l in body:
        m = label.match(l)
        if m:
            cur = m.group(1)
            blocks[cur] = []
            order[cur] = len(order)
            continue
        t = l.strip()
        if not t or t.startswith((";", ".", "//")):
            continue
        blocks[cur].append(t.split()[0] if not t.startswith("s_cbranch") and not t.startswith("s_branch") else t)
    total = collections.Counter()
    for ops in blocks.values():
        for op in ops:
            total[classify(op.split()[0])] += 1
    print(f"{lines[start][:100]}\n  whole kernel: {sum(total.values())} instructions", dict(total))
    if loops:
        names = list(blocks)
        for bi, (b, ops) in enumerate(blocks.items()):
            for op in ops:
                if op.startswith("s_cbranch") or op.startswith("s_branch"):
                    tgt = op.split()[-1]
                    if tgt in order and order[tgt] <= order.get(b, -1):
                        span = names[names.index(tgt):bi + 1]
                        c = collections.Counter()
                        for s in span:
                            for o in blocks[s]:
                                c[classify(o.split()[0])] += 1
                        n = sum(c.values())
                        if n > 200:
                            print(f"  loop {tgt} .. {b}: {n} instructions", dict(c))


main()
