#!/bin/bash
# Round 6, GPU call 13: conv1 as T s on the constant input: tests, timeline, bench with breakdown
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6m
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
( cd "$R" && timeout 1200 python -m pytest tests/test_generator_gpu.py tests/test_render_gpu.py tests/test_layers_gpu.py -q -x -p no:cacheprovider > "$O/pytest.log" 2>&1; echo "pytest rc=$?" >> "$O/pytest.log" )
grep -v "^E  \|^    " "$O/pytest.log" | tail -8
rocprofv3 --kernel-trace -d "$O/trace" -o bench -- python "$R/bench.py" --steps 1 --warmup 1 --batches-per-step 3 --lanes 1 --no-cpu-baseline --no-side-configs --no-breakdown --no-pcie-side > "$O/bench_trace.json" 2> "$O/trace.err"
python "$R/tools/rocpd_timeline.py" "$O/trace/bench_results.db" 2 > "$O/timeline.md"; rm -rf "$O/trace"
head -24 "$O/timeline.md" | cut -c1-110; tail -1 "$O/timeline.md"
cd "$R"
python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-side-configs --no-pcie-side > "$O/bench.json" 2> "$O/bench.err" || tail -5 "$O/bench.err"
python - <<PY
import json
p=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print(round(p["value"],1), "frames/s", p.get("frame_check",{}).get("max_abs_grey_level_diff_graph_vs_eager"))
for r in p["layers"][:10]: print("  %-100s %.4f ms" % (r["name"][:100], r["ms"]))
print(p["roofline"]["kernel"], p["roofline"]["frac"], p.get("whole_forward_executed_frac"))
PY
