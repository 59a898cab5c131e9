#!/usr/bin/env python3
"""Static instruction mix of one kernel split into prologue / main loop (the loop with the most MFMAs) / epilogue.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -Iinclude maua_stylegan2_amd/csrc/modconv_w2d.hip -o /tmp/w2d.s
    python tools/isa_sections.py /tmp/w2d.s modconv_w2d_kernelILi2ELi4ELb0
"""
import re
import sys


def kind(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("v_pk"):
        return "valu_pk"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global", "buffer", "flat", "scratch")):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    lines = open(sys.argv[1]).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and sys.argv[2] in l.split(":")[0])
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    body = lines[start + 1:end]
    labels = {m.group(1): i for i, l in enumerate(body) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}
    loops = []
    for i, l in enumerate(body):
        m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if m and labels.get(m.group(1), len(body)) < i:
            loops.append((labels[m.group(1)], i))

    def count(a, b):
        c = {}
        for l in body[a:b]:
            t = l.strip().split()
            if not t or t[0][0] in ".;" or t[0].endswith(":"):
                continue
            c[kind(t[0])] = c.get(kind(t[0]), 0) + 1
        return dict(sorted(c.items()))

    lo, hi = max(loops, key=lambda ab: sum("v_mfma" in l for l in body[ab[0]:ab[1]]))
    print("kernel", lines[start].split(":")[0])
    print("prologue ", count(0, lo))
    print("main loop", count(lo, hi + 1))
    print("epilogue ", count(hi + 1, len(body)))


main()
