#!/bin/bash
# Round 6, GPU call 1: the style fold — parity of the folded chain / generator, the self-launching bench, then alternating A/B runs of
# the headline region with the fold on and off.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6a
rm -rf "$O"; mkdir -p "$O"
cd "$R"
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/test_layers_gpu.py tests/test_generator_gpu.py tests/test_bench_launch.py -q -m gpu -s -p no:cacheprovider -x > "$O/pytest_fold.log" 2>&1; echo "pytest rc=$?" >> "$O/pytest_fold.log" )
grep -n "style fold\|passed\|failed\|rc=\|Error\|assert" "$O/pytest_fold.log" | tail -40
COMMON="--no-cpu-baseline --no-side-configs --no-pcie-side"
for r in 1 2 3; do
  ( timeout 600 python bench.py --steps 10 $COMMON --no-breakdown > "$O/bench_fold_on_$r.json" 2> "$O/bench_fold_on_$r.err"; echo "on rc=$?" )
  ( timeout 600 python bench.py --steps 10 $COMMON --no-breakdown --no-style-fold > "$O/bench_fold_off_$r.json" 2> "$O/bench_fold_off_$r.err"; echo "off rc=$?" )
done
( timeout 600 python bench.py --steps 10 $COMMON > "$O/bench_layers_on.json" 2> "$O/bench_layers_on.err"; echo "layers on rc=$?" )
( timeout 600 python bench.py --steps 10 $COMMON --no-style-fold > "$O/bench_layers_off.json" 2> "$O/bench_layers_off.err"; echo "layers off rc=$?" )
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/bench_fold_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], round(d["value"], 1), d["frame_check"]["max_abs_grey_level_diff_graph_vs_eager"])
    except Exception as e:
        print(f, "unreadable", e)
rows = {}
for k in ("on", "off"):
    try:
        d = json.loads(open("$O/bench_layers_%s.json" % k).read().strip().splitlines()[-1])
        for r in d["layers"]:
            rows.setdefault(r["name"], {})[k] = r["ms"]
        print(k, "value", round(d["value"], 1), {n: round(v["ms_per_batch"], 3) for n, v in d["conv_kernel_instances"].items()})
    except Exception as e:
        print(k, "unreadable", e)
for n, v in rows.items():
    print(f"{n:60s} on {v.get('on', 0):.4f} off {v.get('off', 0):.4f}")
PY
du -sh "$O"
