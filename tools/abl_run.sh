#!/bin/bash
# Ablation timing of the 2-D Winograd kernel inside ONE gpurun call: tools/abl_run.sh <outdir> <variant> ...  (variants built by
# tools/build_exp.sh <variant> -DMAUA_W2D_ABL=<mask>); two alternating rounds of tools/microbench.py conv fused per variant.
out=$1; shift
mkdir -p $out
for round in 1 2; do
for v in "$@"; do
  python tools/microbench.py conv fused --iters 20 --lib tools/bin/libmaua_$v.so > $out/${v}_$round.json 2> $out/${v}_$round.err
  python - <<PY
import json
p=json.load(open("$out/${v}_$round.json"))
print("%-8s r$round " % "$v" + " ".join("%s %.3f" % (k.replace("plain","p").replace("fused","f"), p[k]["ms"]) for k in p if k.startswith("plain") or k.startswith("fused")))
PY
done
done
