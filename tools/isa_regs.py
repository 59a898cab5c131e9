#!/usr/bin/env python3
"""Registers / scratch / LDS of every kernel in a device assembly listing:
    hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S csrc/<file>.hip -o /tmp/<file>.s && python tools/isa_regs.py /tmp/<file>.s
A spill (scratch > 0) in a K-loop kernel is a regression: tests/test_isa_checks.py holds the product's instances to 0."""
import re
import subprocess
import sys


def demangle(name):
    try:
        return subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
    except FileNotFoundError:
        return name


def kernels(path):
    t = open(path).read()
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", t, re.S):
        body = m.group(2)
        get = lambda key: int(re.search(r"\.amdhsa_" + key + r" (\d+)", body).group(1))  # noqa: E731
        yield demangle(m.group(1)), get("next_free_vgpr"), get("next_free_sgpr"), get("private_segment_fixed_size")


if __name__ == "__main__":
    for name, vgpr, sgpr, scratch in kernels(sys.argv[1]):
        print(f"{name[:110]:110s} vgpr {vgpr:4d} sgpr {sgpr:4d} scratch {scratch}")
