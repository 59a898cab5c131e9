#!/bin/bash
# Round 6, GPU call 12: the whole GPU suite on the tree with the low-resolution entries
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6l
rm -rf "$O"; mkdir -p "$O"
cd "$R"; export TMPDIR=/tmp
( timeout 2700 python -m pytest tests -q -m gpu -p no:cacheprovider > "$O/pytest_gpu.log" 2>&1; echo "pytest rc=$?" >> "$O/pytest_gpu.log" )
grep -v "^E  \|^    " "$O/pytest_gpu.log" | tail -12
