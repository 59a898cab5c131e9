#!/bin/bash
# A/B of FIR kernel builds inside one gpurun call: tools/fir_ab.sh <outdir> <variant> ...  (tools/bin/libmaua_<variant>.so)
out=$1; shift
mkdir -p $out
for round in 1 2 3; do
for v in "$@"; do
  python tools/microbench.py fir --iters 30 --lib tools/bin/libmaua_$v.so > $out/${v}_$round.json 2> $out/${v}_$round.err
  python - <<PY
import json
p=json.load(open("$out/${v}_$round.json"))
print("%-8s r$round fir %.4f ms %.0f GB/s | tail %.4f ms %.0f GB/s" % ("$v", p["fir_1024"]["ms"], p["fir_1024"]["gbs"], p["fir_tail_1024"]["ms"], p["fir_tail_1024"]["gbs"]))
PY
done
done
