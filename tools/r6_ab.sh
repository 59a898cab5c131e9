#!/bin/bash
# alternating A/B of bench.py flags inside one GPU call:  FLAG_B="--no-lowres-fusion" REPS=3 bash tools/r6_ab.sh
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6ab
mkdir -p "$O"
cd "$R"; export TMPDIR=/tmp
for rep in $(seq 1 ${REPS:-3}); do
for flag in "${FLAG_A:-}" "${FLAG_B:---no-lowres-fusion}"; do
python bench.py --steps ${STEPS:-10} --warmup 2 --no-cpu-baseline --no-side-configs --no-pcie-side --no-breakdown $flag > "$O/bench.json" 2> "$O/bench.err" || tail -5 "$O/bench.err"
python - <<PY
import json
p=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("%-28s" % "${flag:-(default)}", round(p["value"],1), "frames/s", p.get("frame_check",{}).get("max_abs_grey_level_diff_graph_vs_eager"))
PY
done; done
