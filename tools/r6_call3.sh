#!/bin/bash
# Round 6, GPU call 3: the whole GPU suite on the current tree, e2e stage times, a lanes sweep with the folded kernels.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6c
rm -rf "$O"; mkdir -p "$O"
cd "$R"
export TMPDIR=/tmp
( timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider > "$O/pytest_gpu.log" 2>&1; echo "pytest rc=$?" >> "$O/pytest_gpu.log" )
tail -5 "$O/pytest_gpu.log"
( timeout 600 python tools/e2e_config3.py --repeat 4 --stage-times > "$O/e2e_config3_stages.txt" 2>&1; echo "e2e rc=$?" )
grep -n "E2E run\|preprocessing took\|rendered\|STAGE load_generator\|STAGE get_noise\|STAGE   gc\|STAGE get_latents\|STAGE generate_latents\|STAGE render\|STAGE initialize" "$O/e2e_config3_stages.txt" | tail -40
for cfg in "8 3" "8 2" "8 4" "8 3" "8 4"; do
  set -- $cfg
  python bench.py --batch $1 --lanes $2 --steps 10 --warmup 2 --no-cpu-baseline --no-side-configs --no-breakdown --no-pcie-side > "$O/b$1_l$2.json" 2> "$O/b$1_l$2.err"
  python - <<PY
import json
try:
    p=json.loads(open("$O/b$1_l$2.json").read().strip().splitlines()[-1])
    print("batch $1 lanes $2:", round(p["value"],1), "frames/s", p.get("frame_check"))
except Exception as e:
    print("batch $1 lanes $2: failed", e)
PY
done
