#!/bin/bash
# `layers` rows of the 4^2 .. 32^2 block for several builds of the library (bench.py --lib): LIBS="a.so b.so" bash tools/r6_layers_ab.sh
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6lab
mkdir -p "$O"
cd "$R"; export TMPDIR=/tmp
for rep in 1 2; do
for lib in $LIBS; do
python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-side-configs --no-pcie-side --lib $lib > "$O/bench.json" 2> "$O/bench.err" || tail -5 "$O/bench.err"
python - <<PY
import json
p=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
rows=p["layers"][:${NROWS:-7}]
print("%-28s %7.1f f/s | " % ("$lib"[-24:], p["value"]) + " ".join("%.1f" % (r["ms"]*1e3) for r in rows))
PY
done; done
