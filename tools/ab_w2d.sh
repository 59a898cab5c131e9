#!/bin/bash
# A/B of two libmaua builds on the w2d layers, alternating runs (clock / thermal drift between runs is a few percent): min of N.
# usage: tools/ab_w2d.sh [libA] [libB]   (defaults: tools/ab/libmaua_prev.so, the in-tree library)
cd "$(dirname "$0")/.."
A=${1:-tools/ab/libmaua_prev.so}; B=${2:-maua_stylegan2_amd/csrc/libmaua_hip.so}
for i in 1 2 3 4; do
  python tools/microbench.py conv fused --iters 20 --wino2d-min-cout 32 --lib $A 2>/dev/null > /tmp/ab_A$i.json
  python tools/microbench.py conv fused --iters 20 --wino2d-min-cout 32 --lib $B 2>/dev/null > /tmp/ab_B$i.json
done
python - <<'PY'
import json
for t in "AB":
    runs = [json.load(open(f"/tmp/ab_{t}{i}.json")) for i in (1, 2, 3, 4)]
    keys = [k for k, v in runs[0].items() if "w2d" in v.get("kernel", "")]
    print(t, {k: (round(min(r[k]["ms"] for r in runs), 3), round(max(r[k]["ms"] for r in runs), 3)) for k in keys})
PY
