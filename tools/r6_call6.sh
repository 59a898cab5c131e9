#!/bin/bash
# Round 6, GPU call 6: timeline of one serial graph replay (kernel trace), the flat-run change of the 8^2 -> 16^2 layer (tests + rows)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6f
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
( cd "$R" && timeout 900 python -m pytest tests/test_layers_gpu.py -q -x -p no:cacheprovider -k "shapes_vs_oracle or upconv" > "$O/pytest_layers.log" 2>&1; echo "pytest rc=$?" >> "$O/pytest_layers.log" )
tail -3 "$O/pytest_layers.log"
rocprofv3 --kernel-trace -d "$O/trace" -o bench -- python "$R/bench.py" --steps 1 --warmup 1 --batches-per-step 3 --lanes 1 --no-cpu-baseline --no-side-configs --no-breakdown --no-pcie-side > "$O/bench_trace.json" 2> "$O/trace.err"
python "$R/tools/rocpd_timeline.py" "$O/trace/bench_results.db" 2 > "$O/timeline.md"; rm -rf "$O/trace"
cat "$O/timeline.md"
( cd "$R" && python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-side-configs --no-pcie-side > "$O/bench.json" 2> "$O/bench.err" )
python - <<PY
import json
p=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print(round(p["value"],1), "frames/s", p.get("frame_check"))
for r in p["layers"][:13]: print("  %-55s %.4f ms" % (r["name"], r["ms"]))
PY
