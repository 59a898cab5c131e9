#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for v in base; do
  lib=""; [ $v != base ] && lib="$R/tools/bin/libmaua_$v.so"
  O=$R/gpurun_out/edge_$v; rm -rf $O
  rocprofv3 --kernel-trace -d $O -o t -- python $R/tools/edge_probe.py $lib > $O.log 2>&1
  echo "== $v"; grep "us per" $O.log; python $R/tools/rocpd_summary.py $(find $O -name "*.db" | head -1) | grep "up2d"
done
