#!/bin/bash
# Fused StyledConv+ToRGB layers: mode 3 vs mode 5 and the ablation of mode 5 (w2d-debug bits: 1 no MFMA, 2 no DMA after the first
# chunk, 4 no feature stores, 8 no epilogue, 16 one K chunk only)
cd "$(dirname "$0")/.."
show() { python - "$1" "$2" <<'PY'
import json, sys
r = json.load(open(sys.argv[2]))
print(sys.argv[1], {k: (round(v["ms"], 3), v["kernel"][8:30]) for k, v in r.items()})
PY
}
python tools/microbench.py fused --iters 20 2>/dev/null > /tmp/f_m3.json; show "mode3      " /tmp/f_m3.json
for d in ${DBG:-0 1 2 3 4}; do
  python tools/microbench.py fused --iters 20 --wino2d-min-cout 32 --w2d-debug $d 2>/dev/null > /tmp/f_$d.json; show "mode5 dbg=$d" /tmp/f_$d.json
done
