#!/bin/bash
# HBM traffic of the standalone upfirdn2d kernels (plain, up = 2, down = 2) from the PMC counters: one pass per counter, kernel trace only.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmc_fir
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --kernel-trace -d "$O/$ctr" -o fir -- python "$R/tools/microbench.py" fir --iters 10 > "$O/$ctr.json" 2> "$O/$ctr.err"
done
cd "$R" && python - <<PY
import sys
sys.path.insert(0, "tools")
import make_profiles as mp
fe = mp.table("$O/FETCH_SIZE/fir_results.db", keep=("fir_",))
wr = mp.table("$O/WRITE_SIZE/fir_results.db", keep=("fir_",))
print("| kernel instance | dispatches | read MB (FETCH_SIZE x 2) | written MB (WRITE_SIZE) | avg us |")
print("|---|---:|---:|---:|---:|")
for n, t in fe.items():
    w = wr.get(n, {})
    print(f"| \`{n}\` | {t['n']} | {2 * t['FETCH_SIZE'] / 1024:.1f} | {w.get('WRITE_SIZE', float('nan')) / 1024:.1f} | {t['us']:.1f} |")
PY
