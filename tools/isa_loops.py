#!/usr/bin/env python3
"""Instruction mix of every loop (backward branch) of one kernel in a hipcc -S listing.
Usage: python tools/isa_loops.py file.s <mangled-kernel-substring>"""
import collections
import re
import sys


def main():
    lines = open(sys.argv[1]).read().split("\n")
    key = sys.argv[2]
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*" + re.escape(key) + r"\S*:", l))
    end = next(i for i in range(start, len(lines)) if ".end_amdhsa_kernel" in lines[i] or lines[i].startswith(".Lfunc_end"))
    body = lines[start:end]
    labels = {}
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = i
    for i, l in enumerate(body):
        m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if not m or m.group(1) not in labels or labels[m.group(1)] >= i:
            continue
        blk = body[labels[m.group(1)]:i + 1]
        c = collections.Counter()
        for x in blk:
            x = x.strip()
            if not x or x[0] in ".;" or x.endswith(":"):
                continue
            op = x.split()[0]
            if op.startswith("v_mfma"):
                c["mfma"] += 1
            elif op.startswith("global_load_lds"):
                c["dma"] += 1
            elif op.startswith(("ds_", "global_", "scratch_", "buffer_", "s_load", "s_waitcnt", "s_nop", "s_barrier")):
                c[op] += 1
            elif op.startswith("s_"):
                c["salu"] += 1
            elif op.startswith("v_accvgpr"):
                c["accvgpr"] += 1
            elif op.startswith("v_"):
                c["valu"] += 1
            else:
                c[op] += 1
        tot = sum(v for k, v in c.items())
        print(m.group(1), "lines", len(blk), "instrs", tot, dict(sorted(c.items(), key=lambda kv: -kv[1])))


main()
