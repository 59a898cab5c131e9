#!/bin/bash
# PMC passes over the serial bench (--lanes 1) for the conv kernels: MFMA-busy / wave cycles / waits, then VALU / LDS activity.
# usage (inside gpurun): bash tools/pmc_conv.sh <outdir> [bench args...]
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/$1; shift
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() {  # name counters...
  name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace -d $O/$name -o bench -- python $R/bench.py --steps 1 --warmup 1 --batches-per-step 3 --lanes 1 \
     --no-cpu-baseline --no-breakdown --no-side-configs $EXTRA > /dev/null 2> $O/$name.err
  python $R/tools/rocpd_pmc.py $(find $O/$name -name "*.db" | head -1) modconv_up2d modconv_w2d modconv_mfma fir_tile > $O/$name.md
  find $O/$name -name "*.db" -size +20M -delete
}
EXTRA="$*"
run sq GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES
run act SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU
run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM
cat $O/sq.md $O/act.md $O/lds.md | grep -v "^| kernel\|^|---" | sort | head -150
