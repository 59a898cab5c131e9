#!/bin/bash
# PMC comparison of modconv_w2d_kernel builds on the microbench layers (cycles, MFMA-busy, effective clock)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for tag in prev new; do
  lib=""; [ $tag = prev ] && lib="--lib $R/tools/ab/libmaua_prev.so"
  O=$R/gpurun_out/pmc_w2d_$tag; rm -rf $O
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --kernel-trace -d $O -o mb -- \
    python $R/tools/microbench.py conv --iters 5 --wino2d-min-cout 32 $lib > /dev/null 2> $O.err
  echo "== $tag"; python $R/tools/rocpd_pmc.py $(find $O -name "*.db" | head -1) w2d_kernel | grep -v "^|---"
done
