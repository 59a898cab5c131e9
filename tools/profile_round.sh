#!/bin/bash
# Round-end evidence run on the GPU box (invoked through gpurun): parity suite, smoke, bench (+CPU baseline), rocprofv3
# kernel trace of the bench command (default three-lane run and the strictly serial --lanes 1 run) and the PMC passes
# (each counter group in its own pass, no sys/hip/hsa trace domains).  Outputs land in gpurun_out/prof_final/.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_final
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
( cd "$R" && python -m pytest tests -q -m gpu > "$O/pytest_gpu.log" 2>&1; echo "pytest rc=$?" >> "$O/pytest_gpu.log" )
( cd "$R" && python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > "$O/smoke.log" 2>&1 )
( cd "$R" && python bench.py --steps 20 > "$O/bench_default.json" 2> "$O/bench_default.err" )
# short traced / counted runs: 1 warm-up + 1 timed step of 3 batches (every kernel instance of the path is dispatched several times)
SHORT="--steps 1 --warmup 1 --batches-per-step 3 --no-cpu-baseline --no-side-configs"
rocprofv3 --kernel-trace --stats -d "$O/trace_lanes3" -o bench -- python "$R/bench.py" $SHORT > "$O/bench_trace_lanes3.json" 2> "$O/trace_lanes3.err"
rocprofv3 --kernel-trace --stats -d "$O/trace_lanes1" -o bench -- python "$R/bench.py" $SHORT --lanes 1 > "$O/bench_trace_lanes1.json" 2> "$O/trace_lanes1.err"
# copy accounting: the same traced run with 3 and with 12 batches per step and nothing else in the process (no breakdown, no warm-up);
# whatever does not grow with the batch count is set-up, not per-batch work (make_profiles.py tabulates the copy kernels)
for nb in 3 12; do
  rocprofv3 --kernel-trace -d "$O/trace_nb$nb" -o bench -- python "$R/bench.py" --steps 1 --warmup 0 --batches-per-step $nb --lanes 1 --no-cpu-baseline --no-side-configs --no-breakdown --no-pcie-side > /dev/null 2> "$O/trace_nb$nb.err"
  python "$R/tools/rocpd_summary.py" "$O/trace_nb$nb/bench_results.db" > "$O/trace_nb$nb.md"; rm -rf "$O/trace_nb$nb"   # (the merge-back is capped at 64 MiB)
done
for l in 1 3; do  # (per-kernel summaries are made here: the merge-back is capped at 64 MiB and a trace database is ~10 MiB)
  python "$R/tools/rocpd_summary.py" "$O/trace_lanes$l/bench_results.db" > "$O/trace_lanes$l.md"; rm -rf "$O/trace_lanes$l"
done
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --kernel-trace -d "$O/pmc_$ctr" -o bench -- python "$R/bench.py" $SHORT --lanes 1 --no-breakdown > /dev/null 2> "$O/pmc_$ctr.err"
done
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --kernel-trace -d "$O/pmc_sq" -o bench -- python "$R/bench.py" $SHORT --lanes 1 --no-breakdown > /dev/null 2> "$O/pmc_sq.err"
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU SQ_WAIT_ANY --kernel-trace -d "$O/pmc_lds" -o bench -- python "$R/bench.py" $SHORT --lanes 1 --no-breakdown > /dev/null 2> "$O/pmc_lds.err"
# the standalone upfirdn2d leg needs the breakdown-free run above to contain fir_tile_kernel<4, 4, 4, false>: it does (roofline_upfirdn2d is always timed)
find "$O" -name "*.db" -size +20M -delete   # keep the merge-back under gpurun's 64 MiB limit
du -sh "$O"; tail -2 "$O/pytest_gpu.log"; cat "$O/smoke.log" | tail -1; head -c 600 "$O/bench_default.json"
