#!/bin/bash
# Round 6, closing evidence call of the third session: the bench of record (roofline.traffic from the round's PMC table), sanitizer legs on the
# rebuilt host-ASAN library (ABI 5, with the low-resolution cases in the torch-free driver), the self-launched multi-rank runs, end-to-end config 3.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"
bash tools/r6_final.sh
O=$R/gpurun_out/r6_final
mkdir -p $R/gpurun_out/prof_final
cp "$O/bench_default.json" $R/gpurun_out/prof_final/bench_default.json
cp "$O/bench_default.err" $R/gpurun_out/prof_final/bench_default.err
