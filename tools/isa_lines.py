#!/usr/bin/env python3
"""Static per-source-line issue-cycle profile of one kernel instantiation (no GPU needed).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -gline-tables-only -S --cuda-device-only -Iinclude \\
          maua_stylegan2_amd/csrc/modconv.hip -o /tmp/modconv_g.s
    python tools/isa_lines.py /tmp/modconv_g.s maua_stylegan2_amd/csrc/modconv.hip 'ILi32ELi128ELi1ELi3ELb0ELb1ELi3E' [top]

Every VALU instruction of the kernel counts 4 issue cycles, every SALU 1 (the issue model of profiles/r01_pmc_modconv.md),
attributed to the source line of its .loc; MFMAs are listed separately (64 cycles each).  Instructions are split by PROGRAM
ORDER into "before the first MFMA" / "between" / "after the last MFMA"; the compiler rotates loops, so pieces of the K-chunk
loop also show up in the outer groups (lines of mfma_chunk / issue_patch) — read the groups as code layout, not as phases.  A stretch without its own .loc inherits the
previous line, so one line can collect a whole inlined region."""
import collections
import re
import sys

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from isa_mix import classify  # noqa: E402


def main():
    asm, src_path, pattern = sys.argv[1:4]
    top = int(sys.argv[4]) if len(sys.argv) > 4 else 25
    text = open(asm).read()
    src = open(src_path).read().splitlines()
    m = re.search(r"^(_ZN[^\s:]*modconv_mfma_kernel" + re.escape(pattern) + r"[^\s:]*):", text, flags=re.M)
    if not m:
        sys.exit(f"no kernel matching {pattern}")
    body = text[m.start(): text.find(".Lfunc_end", m.start())].splitlines()
    mf = [i for i, ln in enumerate(body) if re.match(r"^\s+v_mfma", ln)]
    groups = [collections.Counter() for _ in range(3)]
    line = 0
    for i, ln in enumerate(body):
        lm = re.match(r"^\s+\.loc\s+\d+\s+(\d+)", ln)
        if lm:
            line = int(lm.group(1))
            continue
        om = re.match(r"^\s+([a-z_0-9]+)", ln)
        if not om or ln.strip().startswith((".", ";")):
            continue
        kind = classify(om.group(1))
        if kind not in ("valu", "salu"):
            continue
        grp = 0 if i < mf[0] else (2 if i > mf[-1] else 1)
        groups[grp][line] += 4 if kind == "valu" else 1
    print(f"{m.group(1)}\n{len(mf)} MFMA instructions = {64 * len(mf)} cycles per unrolled pass")
    for title, c in zip(("before the first MFMA", "between first and last MFMA", "after the last MFMA"), groups):
        total = sum(c.values())
        print(f"\n{title}: {total} VALU/SALU issue cycles")
        for ln_no, cy in c.most_common(top):
            text_ = src[ln_no - 1].strip()[:100] if 0 < ln_no <= len(src) else ""
            print(f"  {cy:6d} {100 * cy / max(total, 1):5.1f}%  L{ln_no}: {text_}")


if __name__ == "__main__":
    main()
