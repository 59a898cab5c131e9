#!/bin/bash
# Does fp32 VALU work overlap fp32 MFMA work?  PMC passes over the bench command (serial lanes): matrix-pipe busy cycles, cycles in
# which VALU and MFMA execute together, VALU / VMEM / LDS / scalar instruction-active time.  Output: gpurun_out/pmc_coexec/*.md
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmc_coexec; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 2 --warmup 1 --lanes 1 --no-cpu-baseline --no-breakdown"
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES --kernel-trace -d $O/p1 -o b -- $CMD > /dev/null 2> $O/p1.err
rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES --kernel-trace -d $O/p2 -o b -- $CMD > /dev/null 2> $O/p2.err
for p in p1 p2; do python $R/tools/rocpd_pmc.py $(find $O/$p -name "*.db" | head -1) modconv > $O/$p.md; done
find $O -name "*.db" -delete
cat $O/p1.md | head -60
