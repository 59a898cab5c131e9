#!/bin/bash
# A/B of library variants inside ONE gpurun call: tools/ab_bench.sh <outdir> <lib or "base"> ... — alternating bench runs
# (steps 12, no CPU baseline / side configs), prints frames/s and the per-layer times of the families under study.
out=$1; shift
mkdir -p $out
for round in 1 2; do
for v in "$@"; do
  if [ "$v" = base ]; then lib=""; else lib="--lib tools/bin/libmaua_$v.so"; fi
  python bench.py --steps 12 --no-cpu-baseline --no-side-configs $lib > $out/bench_${v}_$round.json 2> $out/bench_${v}_$round.err
  python - <<PY
import json
p=json.load(open("$out/bench_${v}_$round.json"))
fam=p["kernel_families"]
print("$v", "round $round", "fps %.1f" % p["value"], "check", p["frame_check"]["max_abs_grey_level_diff_graph_vs_eager"],
      " ".join("%s %.3f" % (k, fam[k]["ms_isolated"]) for k in fam))
print("   up:", " ".join("%.4f" % l["ms"] for l in p["layers"] if "upconv" in l["name"]), "| plain:", " ".join("%.4f" % l["ms"] for l in p["layers"] if l["name"].startswith("convs.") and "+" in l["name"]))
PY
done
done
