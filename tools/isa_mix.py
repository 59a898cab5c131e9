#!/usr/bin/env python3
"""Static instruction mix of the hottest loop of every modconv_mfma_kernel instantiation, from the device assembly:

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -Iinclude maua_stylegan2_amd/csrc/modconv.hip -o /tmp/modconv.s
    python tools/isa_mix.py /tmp/modconv.s [--all] > profiles/rNN_isa_modconv.md

For each kernel: VGPR count / spills from the metadata, and for the loop (backward branch range) that contains the most
MFMAs — the K-chunk loop — the number of MFMA, other VALU, SALU, LDS reads / writes, global / DMA loads, waits and barriers
per iteration.  Under the measured issue model of this chip (cycles ~ 64 x MFMA + 4 x VALU + SALU per wave,
profiles/r01_pmc_modconv.md) the last column estimates the share of issue cycles the MFMAs can have at best."""
import re
import sys


def classify(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_read") or op.startswith("ds_load"):
        return "lds_rd"
    if op.startswith("ds_"):
        return "lds_wr"
    if op.startswith("global_load_lds") or (op.startswith("global_load") and "lds" in op):
        return "dma"
    if op.startswith(("global_load", "buffer_load", "flat_load", "scratch_load")):
        return "vmem_ld"
    if op.startswith(("global_store", "buffer_store", "flat_store", "scratch_store")):
        return "vmem_st"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path = sys.argv[1]
    show_all = "--all" in sys.argv
    text = open(path).read()
    meta = {}
    for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)", text):
        meta[m.group(1)] = (int(m.group(2)), int(m.group(3)))
    kernels = [(m.group(1), m.start()) for m in re.finditer(r"^(_ZN[^\s:]*modconv_mfma_kernel[^\s:]*):", text, flags=re.M)]
    print("| BM | BN | WM | MODE | MULTI | FAST | MAXP | VGPR | spill | loop: MFMA | VALU | SALU | LDS rd | LDS wr | DMA | vmem ld | wait | barrier | MFMA share of issue |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    for idx, (name, start) in enumerate(kernels):
        end = text.find(".Lfunc_end", start)
        body = text[start:end].splitlines()
        t = re.search(r"ILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELb(\d)ELb(\d)ELi(\d+)E", name)
        bm, bn, wm, mode, multi, fast, maxp = [int(v) for v in t.groups()]
        if not show_all and not (fast and not multi):
            continue
        labels = {}
        for i, ln in enumerate(body):
            lm = re.match(r"^(\.LBB\d+_\d+):", ln)
            if lm:
                labels[lm.group(1)] = i
        best = None
        for i, ln in enumerate(body):
            bm_ = re.match(r"^\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)", ln) or re.match(r"^\s+s_branch\s+(\.LBB\d+_\d+)", ln)
            if bm_ and bm_.group(1) in labels and labels[bm_.group(1)] < i:
                lo = labels[bm_.group(1)]
                counts = {}
                for ln2 in body[lo:i + 1]:
                    om = re.match(r"^\s+([a-z_0-9]+)", ln2)
                    if om and not ln2.strip().startswith((".", ";")):
                        c = classify(om.group(1))
                        counts[c] = counts.get(c, 0) + 1
                if best is None or counts.get("mfma", 0) > best.get("mfma", 0):
                    best = counts
        best = best or {}
        g = lambda k: best.get(k, 0)  # noqa: E731
        issue = 64 * g("mfma") + 4 * g("valu") + g("salu") + g("wait") + g("barrier") + 4 * (g("lds_rd") + g("lds_wr") + g("dma") + g("vmem_ld"))
        share = 64 * g("mfma") / issue if issue else 0.0
        vg, sp = meta.get(name, (0, 0))
        print(f"| {bm} | {bn} | {wm} | {mode} | {multi} | {fast} | {maxp} | {vg} | {sp} | {g('mfma')} | {g('valu')} | {g('salu')} | "
              f"{g('lds_rd')} | {g('lds_wr')} | {g('dma')} | {g('vmem_ld')} | {g('wait')} | {g('barrier')} | {share:.2f} |")


if __name__ == "__main__":
    main()
