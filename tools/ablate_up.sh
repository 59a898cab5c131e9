#!/bin/bash
# Ablation of the transposed-conv kernels (modes 1 / 4): --conv-debug 32 = no weight DMA after the first chunk,
# 64 = no patch DMA, 96 = neither, 2 = no MFMA.  ms per launch, batch 8.
cd "$(dirname "$0")/.."
for d in 0 32 64 96 2; do
  python tools/microbench.py conv --iters 20 --conv-debug $d 2>/dev/null > /tmp/ab_$d.json
  python - "$d" <<'PY'
import json, sys
d = sys.argv[1]
r = json.load(open(f"/tmp/ab_{d}.json"))
print(d, {k: round(v["ms"], 3) for k, v in r.items() if k.startswith("up")})
PY
done
