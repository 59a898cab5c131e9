#!/bin/bash
# quick check: selected tests + the first dispatches of one serial replay
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6q
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
( cd "$R" && timeout 900 python -m pytest ${TESTS:-tests/test_generator_gpu.py} -q -x -p no:cacheprovider -k "${KEXPR:-const_conv or golden}" > "$O/pytest.log" 2>&1; echo "pytest rc=$?" >> "$O/pytest.log" )
grep -v "^E  \|^    " "$O/pytest.log" | tail -6
rocprofv3 --kernel-trace -d "$O/trace" -o bench -- python "$R/bench.py" --steps 1 --warmup 1 --batches-per-step 3 --lanes 1 --no-cpu-baseline --no-side-configs --no-breakdown --no-pcie-side > "$O/bench_trace.json" 2> "$O/trace.err"
python "$R/tools/rocpd_timeline.py" "$O/trace/bench_results.db" 2 > "$O/timeline.md"; rm -rf "$O/trace"
head -${NLINES:-24} "$O/timeline.md" | cut -c1-110; tail -1 "$O/timeline.md"
