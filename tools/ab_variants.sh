#!/bin/bash
# A/B of the prepared modconv experiments (profiles/r01_isa_modconv.md) in ONE GPU call.
#
#   here (CPU, ~2 min per variant):   bash tools/ab_variants.sh build
#   on the GPU box:                   gpurun --timeout 900 -- 'bash tools/ab_variants.sh run'
#
# build: tools/bin/libmaua_{buf,pipebuf,tappipebuf}.so (they travel with the gpurun snapshot).
# run:   per variant — conv micro-benchmark (30-launch averages per layer), the layer / property / generator parity suites
#        against the variant library (MAUA_TEST_LIB), the NaN-poison soak; then bench.py stays on the default library.
#        Results under gpurun_out/ab/.
set -u
cd "$(dirname "$0")/.."
# name:flag,flag ... — round 1's three switches (buffer-form DMA, F(4,3) pipeline, tap look-ahead) were timed with this script
# (profiles/r02_ab_variants.md) and became the default schedule; list new experiment macros here or in $VARIANTS.
VARIANTS="${VARIANTS:-}"
case "${1:-}" in
build)
    for v in $VARIANTS; do
        name=${v%%:*}; flags=${v#*:}
        bash tools/build_exp.sh "$name" ${flags//,/ } || exit 1
    done
    ;;
run)
    O=gpurun_out/ab; mkdir -p $O
    python tools/microbench.py conv --iters 30 > $O/conv_base.json 2> $O/conv_base.err
    for v in $VARIANTS; do
        name=${v%%:*}; lib=tools/bin/libmaua_$name.so
        [ -f "$lib" ] || { echo "missing $lib (run: bash tools/ab_variants.sh build)"; continue; }
        python tools/microbench.py conv --iters 30 --lib $lib > $O/conv_$name.json 2> $O/conv_$name.err
        MAUA_TEST_LIB=$lib timeout 300 python -m pytest tests/test_layers_gpu.py tests/test_property_gpu.py \
            tests/test_generator_gpu.py -q -m gpu -x > $O/pytest_$name.log 2>&1
        echo "$name: pytest rc=$? $(tail -1 $O/pytest_$name.log)"
    done
    python - <<'PY'
import glob, json, os
base = json.load(open("gpurun_out/ab/conv_base.json"))
for path in sorted(glob.glob("gpurun_out/ab/conv_*.json")):
    name = os.path.basename(path)[5:-5]
    if name == "base":
        continue
    try:
        var = json.load(open(path))
    except Exception as e:  # a variant that crashed leaves an empty file
        print(name, "unreadable:", e)
        continue
    rows = []
    for key, b in base.items():
        v = var.get(key)
        if isinstance(b, dict) and isinstance(v, dict) and "ms" in b and "ms" in v:
            rows.append((key, b["ms"], v["ms"]))
    tb, tv = sum(r[1] for r in rows), sum(r[2] for r in rows)
    print(f"== {name}: sum of layer launches {tb:.3f} -> {tv:.3f} ms ({100 * (tv / tb - 1):+.1f} %)")
    for key, b, v in rows:
        if abs(v / b - 1) > 0.01:
            print(f"   {key:40s} {b:8.4f} -> {v:8.4f} ms ({100 * (v / b - 1):+.1f} %)")
PY
    ;;
*)
    echo "usage: $0 build|run"; exit 2 ;;
esac
