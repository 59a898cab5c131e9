#!/bin/bash
# Round 6, GPU call 4: marginal cost of layer groups under the lanes (tools/ablate_groups.py), sanity run of the tree.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6d
rm -rf "$O"; mkdir -p "$O"
cd "$R"
export TMPDIR=/tmp
timeout 900 python tools/ablate_groups.py --lanes 3 --groups none,small,tails,fused,w2dw,convs.13,mid,convs.5,none > "$O/ablate_l3.txt" 2> "$O/ablate_l3.err"; echo "rc=$?"
cat "$O/ablate_l3.txt"; tail -3 "$O/ablate_l3.err"
timeout 600 python tools/ablate_groups.py --lanes 1 --groups none,small,tails,none > "$O/ablate_l1.txt" 2> "$O/ablate_l1.err"; echo "rc=$?"
cat "$O/ablate_l1.txt"; tail -3 "$O/ablate_l1.err"
