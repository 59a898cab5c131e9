#!/bin/bash
# Round 6, GPU call 7: the ToRGB chain as a side branch of the captured forward: parity tests + A/B at 3 lanes and 1 lane
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6g
rm -rf "$O"; mkdir -p "$O"
cd "$R"; export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/test_generator_gpu.py tests/test_render_gpu.py -q -x -p no:cacheprovider > "$O/pytest.log" 2>&1; echo "pytest rc=$?" >> "$O/pytest.log" )
tail -5 "$O/pytest.log"
B="--steps 8 --warmup 2 --no-cpu-baseline --no-side-configs --no-pcie-side --no-breakdown"
for rep in 1 2; do
for cfg in "3 " "3 --no-rgb-side-branch" "1 " "1 --no-rgb-side-branch"; do
  set -- $cfg
  python bench.py $B --lanes $1 ${2:-} > "$O/b.json" 2> "$O/b.err" || tail -3 "$O/b.err"
  python - <<PY
import json
p=json.loads(open("$O/b.json").read().strip().splitlines()[-1])
print("lanes $1 ${2:-side-branch}:", round(p["value"],1), "frames/s", p.get("frame_check",{}).get("max_abs_grey_level_diff_graph_vs_eager"))
PY
done; done
