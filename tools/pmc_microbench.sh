#!/bin/bash
# PMC passes over tools/microbench.py conv (inside gpurun): tools/pmc_microbench.sh <outdir> <kernel-name filter> [microbench args...]
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/$1; F=$2; shift 2
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() {
  name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace -d $O/$name -o mb -- python $R/tools/microbench.py conv --iters 3 $EXTRA > /dev/null 2> $O/$name.err
  python $R/tools/rocpd_pmc.py $(find $O/$name -name "*.db" | head -1) $F > $O/$name.md
  find $O/$name -name "*.db" -delete
}
EXTRA="$*"
run sq GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES
run act SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM
run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_MFMA
cat $O/sq.md $O/act.md $O/lds.md | grep -v "^| kernel\|^|---" | sort | head -80
