#!/usr/bin/env python3
"""Turn gpurun_out/prof_final/ (written by tools/profile_round.sh on the GPU box) into the tracked summaries under profiles/.

    python tools/make_profiles.py [round-tag, default r01]
"""
import json
import re
import shutil
import sqlite3
import sys

O = "gpurun_out/prof_final"
TAG = sys.argv[1] if len(sys.argv) > 1 else "r06"


def table(db, keep=("modconv_mfma", "modconv_w2d", "modconv_up2d", "up2d_edge", "up2d_seam", "fir_", "torgb", "frames_to_u8", "style_affine", "demod_kernel",
                     "reduce_tail")):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection "
                       "group by kernel_name, counter_name").fetchall()
    tab = {}
    for n, c, k, v, dur in rows:
        n = re.sub(r"\(anonymous namespace\)::", "", n)
        n = re.sub(r"\(.*", "", n).replace("void ", "")
        if not any(s in n for s in keep):
            continue
        tab.setdefault(n, {})[c] = v
        tab[n]["n"] = k
        tab[n]["us"] = dur / 1e3
    return tab


def table_by_grid(db, counter):
    """{instance: {grid_size (threads): {counter: avg, n, us}}} — for instances launched more than once per forward with different grids
    (the fused up-sampling layers share one instance)."""
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select kernel_name, grid_size, count(*), avg(value), avg(duration) from counters_collection where counter_name = ? "
                       "group by kernel_name, grid_size", (counter,)).fetchall()
    tab = {}
    for n, grid, k, v, dur in rows:
        n = re.sub(r"\(anonymous namespace\)::", "", n)
        n = re.sub(r"\(.*", "", n).replace("void ", "")
        tab.setdefault(n, {})[int(grid)] = {counter: v, "n": k, "us": dur / 1e3}
    return tab


def last_json(path):
    return json.loads(open(path).read().strip().splitlines()[-1])


def main():
    shutil.copy(f"{O}/pytest_gpu.log", f"profiles/{TAG}_pytest_gpu.log")
    for src, dst in (("bench_default", "bench_default"), ("bench_trace_lanes1", "bench_rocprof_lanes1"),
                     ("bench_trace_lanes3", "bench_rocprof_lanes3")):
        shutil.copy(f"{O}/{src}.json", f"profiles/{TAG}_{dst}.json")
    l1, l2 = last_json(f"{O}/bench_trace_lanes1.json"), last_json(f"{O}/bench_trace_lanes3.json")

    def stats(md):
        return "\n".join(line for line in open(md).read().splitlines()[1:] if "at::native" not in line)

    def copies():
        rows = {}
        for nb in (3, 12):
            for line in open(f"{O}/trace_nb{nb}.md"):
                cells = [c.strip() for c in line.split("|")]
                if len(cells) < 4 or not cells[2].isdigit():
                    continue
                name, n = cells[1], int(cells[2])
                key = ("copy kernels (`__amd_rocclr_copyBuffer*`, `__amd_rocclr_fillBuffer*`)" if "__amd_rocclr" in name else
                       "`modconv_up2d_kernel` (the 5 mode-6 layers of a batch)" if "modconv_up2d_kernel" in name else None)
                if key:
                    rows.setdefault(key, {}).setdefault(nb, 0)
                    rows[key][nb] += n
        out = ["`rocprofv3 --kernel-trace -- python bench.py --steps 1 --warmup 0 --batches-per-step N --lanes 1 --no-cpu-baseline --no-side-configs",
               "--no-breakdown --no-pcie-side` with N = 3 and N = 12 (set-up, capture and the frame check against the eager forward are in both",
               "runs; only the number of replayed batches differs):", "",
               "| dispatches of | N = 3 | N = 12 | per extra batch |", "|---|---:|---:|---:|"]
        for key, v in rows.items():
            out.append(f"| {key} | {v.get(3, 0)} | {v.get(12, 0)} | {(v.get(12, 0) - v.get(3, 0)) / 9:.2f} |")
        out.append("")
        out.append("The per-frame inputs reach the kernels through the device-side frame source (`maua_frame_source_seek` = one 4-byte "
                   "`hipMemsetD32Async` node per replay, which is what the fill kernel row counts); no input is copied per batch.")
        return "\n".join(out)

    with open(f"profiles/{TAG}_bench_kernel_stats.md", "w") as f:
        f.write(f"""# rocprofv3 --kernel-trace --stats of bench.py ({TAG}, final kernels of the round)

Commands (tools/profile_round.sh, run through gpurun on one MI355X):

    rocprofv3 --kernel-trace --stats -- python bench.py --steps 1 --warmup 1 --batches-per-step 3 --no-cpu-baseline --no-side-configs --lanes 1   # strictly serial batches
    rocprofv3 --kernel-trace --stats -- python bench.py --steps 1 --warmup 1 --batches-per-step 3 --no-cpu-baseline --no-side-configs             # default: 3 graph lanes

The serial run is the one whose per-kernel averages are comparable with the live HIP-event numbers in the bench JSON
(profiles/{TAG}_bench_rocprof_lanes1.json: {l1['value']:.0f} frames/s, roofline.launch_ms {l1['roofline']['launch_ms']:.3f} ms for
{l1['roofline']['kernel']}).  Under three lanes ({l2['value']:.0f} frames/s) the kernels of consecutive batches share the device, so
individual durations stretch while the step time drops.  Each trace also contains bench.py's per-layer breakdown pass (eager
launches), which is why calls != steps x layers.  Template arguments of modconv_mfma_kernel: <BM, BN, WM, MODE, MULTI, FAST, MAXP>,
MODE 0 direct, 1 transposed (polyphase), 2 Winograd F(2,3), 3 Winograd F(4,3), 4 transposed with F(2,2) on the even x-phase;
modconv_w2d_kernel<TM, TN, MINB, PRE> = 2-D Winograd F(2x4,3x3) (mode 5), modconv_w2dw_kernel<PRE> = its wave-complete form for 32 output channels; modconv_up2d_kernel<CC, FUSE, PRE, TW> = transposed with F(2,2) on both axes (mode 6; FUSE = 2: + blur + noise + bias + act in the same kernel; TW = 16: the 16 x 16-position tiles of the low-resolution entry, K split into slabs),
up2d_edge_kernel = its two edge lines.  PRE = true: the instance for an input map that arrives multiplied by the layer's styles (the style fold, round 6): no style multiplies in its K loop.  (These short runs time one cold step: the frames/s quoted here are not the headline.)

## Copies per batch

{copies()}

## --lanes 1
{stats(f'{O}/trace_lanes1.md')}

## default (3 lanes)
{stats(f'{O}/trace_lanes3.md')}
""")

    sq, lds = table(f"{O}/pmc_sq/bench_results.db"), table(f"{O}/pmc_lds/bench_results.db")
    fe, wr = table(f"{O}/pmc_FETCH_SIZE/bench_results.db"), table(f"{O}/pmc_WRITE_SIZE/bench_results.db")
    lines = [f"""
## Final kernels of {TAG}, measured inside bench.py

Command (tools/profile_round.sh): `rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY
SQ_WAVE_CYCLES --kernel-trace -- python bench.py --steps 1 --warmup 1 --batches-per-step 3 --lanes 1 --no-cpu-baseline
--no-side-configs --no-breakdown` and a second pass with `--pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32
SQ_INSTS_VALU SQ_WAIT_ANY` (counters only, with --kernel-trace; no sys/hip/hsa trace domains).  Averages over every dispatch of the template instance in the run (batch 8, 1024^2 generator).
MFMA busy % = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs); clock = (GRBM_GUI_ACTIVE / 8) / duration.
Template arguments: <BM, BN, WM, MODE (0 direct, 1 transposed, 2 Winograd F(2,3), 3 Winograd F(4,3), 4 transposed + F(2,2)),
MULTI, FAST, MAXP>; modconv_w2d_kernel<TM, TN, MINB, PRE> / modconv_w2dw_kernel<PRE> = mode 5, modconv_up2d_kernel<CC, FUSE, PRE, TW> = mode 6 (PRE: pre-scaled input, the style fold; TW: position columns of a tile).  "executed TFLOP/s" = MFMA_MOPS_F32 x 512 flop /
duration; "non-MFMA VALU per MFMA" = (SQ_INSTS_VALU - MOPS / 4) / (MOPS / 4).

| kernel instance | dispatches | avg us | clock GHz | MFMA busy % | executed TFLOP/s | non-MFMA VALU per MFMA | wave cycles waiting on an instruction % | waiting on a counter / barrier % | LDS bank conflicts % of LDS active |
|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|"""]
    for n, t in sorted(sq.items(), key=lambda kv: -kv[1]["us"] * kv[1]["n"]):
        if "modconv" not in n:
            continue
        cyc = t["GRBM_GUI_ACTIVE"] / 8
        ll = lds.get(n, {})
        mops = ll.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0.0)
        mfmas = max(mops / 4.0, 1.0)
        lines.append(f"| `{n}` | {t['n']} | {t['us']:.0f} | {cyc / t['us'] / 1e3:.2f} | "
                     f"{t['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024) * 100:.1f} | "
                     f"{mops * 512 / (ll.get('us', t['us']) * 1e-6) / 1e12:.1f} | "
                     f"{(ll.get('SQ_INSTS_VALU', 0.0) - mfmas) / mfmas:.2f} | "
                     f"{t['SQ_WAIT_INST_ANY'] / t['SQ_WAVE_CYCLES'] * 100:.0f} | "
                     f"{100 * ll.get('SQ_WAIT_ANY', 0) / max(t['SQ_WAVE_CYCLES'], 1):.0f} | "
                     f"{100 * ll.get('SQ_LDS_BANK_CONFLICT', 0) / max(ll.get('SQ_LDS_IDX_ACTIVE', 1), 1):.1f} |")
    open(f"profiles/{TAG}_pmc_modconv.md", "w").write(f"# PMC counters of the modulated-conv kernels ({TAG})\n" + "\n".join(lines) + "\n")

    tr = [f"# HBM traffic per launch inside bench.py ({TAG}; rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes)", "",
          "Command: `rocprofv3 --pmc <CTR> --kernel-trace -- python bench.py --steps 1 --warmup 1 --batches-per-step 3 --lanes 1 --no-cpu-baseline --no-side-configs --no-breakdown`",
          "(tools/profile_round.sh).  Units: KB per dispatch, averaged over the dispatches of one template instance.  Corrections as calibrated",
          "in r01_pmc_upfirdn2d.md against kernels of known traffic (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports exactly 1/2 of",
          "the bytes read -> x2; WRITE_SIZE is exact.", "",
          "| kernel instance | dispatches | FETCH_SIZE KB | read KB (x2) | WRITE_SIZE KB | avg us |", "|---|---:|---:|---:|---:|---:|"]
    for n, t in sorted(fe.items(), key=lambda kv: -kv[1]["us"] * kv[1]["n"]):
        w = wr.get(n, {})
        tr.append(f"| `{n}` | {t['n']} | {t['FETCH_SIZE']:.0f} | {2 * t['FETCH_SIZE']:.0f} | {w.get('WRITE_SIZE', float('nan')):.0f} | {t['us']:.0f} |")
    open(f"profiles/{TAG}_pmc_traffic.md", "w").write("\n".join(tr) + "\n")
    # machine-readable twin, read by bench.py for roofline.traffic (no literals in bench.py): bytes per dispatch, averaged over
    # the dispatches of an instance in the short serial run.  Forwards executed in that run = dispatches of a once-per-forward
    # kernel (eager warm-up of the capture, every replay of the warm-up / timed / PCIe-inclusive regions, the frame check);
    # "dispatches_per_step" = dispatches per FORWARD (= per batch)
    steps_in_run = int(next(t["n"] for n, t in fe.items() if n.startswith("style_affine_kernel")))
    kernels = {}
    fe_grid = table_by_grid(f"{O}/pmc_FETCH_SIZE/bench_results.db", "FETCH_SIZE")
    wr_grid = table_by_grid(f"{O}/pmc_WRITE_SIZE/bench_results.db", "WRITE_SIZE")
    for n, t in fe.items():
        w = wr.get(n, {})
        if "WRITE_SIZE" not in w:
            continue
        rec = {"read_bytes": 2.0 * t["FETCH_SIZE"] * 1024.0, "write_bytes": w["WRITE_SIZE"] * 1024.0, "dispatches": t["n"],
               "avg_us": t["us"], "dispatches_per_step": t["n"] / steps_in_run if t["n"] % steps_in_run == 0 else None}
        if rec["dispatches_per_step"] != 1 and len(fe_grid.get(n, {})) > 1:
            # several launches of one instance per forward: the same averages per launch GRID (threads), so that one launch can be told apart
            rec["by_grid"] = {str(g): {"read_bytes": 2.0 * a["FETCH_SIZE"] * 1024.0, "write_bytes": wr_grid[n][g]["WRITE_SIZE"] * 1024.0,
                                       "dispatches": a["n"], "avg_us": a["us"],
                                       "dispatches_per_step": a["n"] / steps_in_run if a["n"] % steps_in_run == 0 else None}
                              for g, a in fe_grid[n].items() if g in wr_grid.get(n, {})}
        if n.startswith("fir_strip_kernel") or n in ("fir_tile_kernel<4, 4, 4, false>", "fir_tile_kernel<4, 4, 4, false, 24>"):  # (the plain op: only bench.py's standalone leg launches it)
            rec["planes"] = 256  # bench.py's standalone upfirdn2d leg: [8, 32, 1025, 1025]
        kernels[n] = rec
    bench = last_json(f"{O}/bench_default.json")
    with open(f"profiles/{TAG}_pmc_traffic.json", "w") as f:
        json.dump({"source": "tools/profile_round.sh: rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE --kernel-trace -- python bench.py "
                             "--steps 1 --warmup 1 --batches-per-step 3 --lanes 1 --no-cpu-baseline --no-side-configs --no-breakdown",
                   "correction": "FETCH_SIZE x2 (reports 1/2 of the bytes read on gfx950, calibrated on a known-size copy), WRITE_SIZE exact",
                   "batch": bench["config"]["batch"], "size": 1024, "kernels": kernels}, f, indent=1)
    print(f"wrote profiles/{TAG}_*")


if __name__ == "__main__":
    main()
