set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/w2d; rm -rf $O; mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_layers_gpu.py -m gpu -x -q -k "winograd2d or fused_torgb" 2>&1 | tail -3
for dbg in 0 1 2 3 4; do python tools/microbench.py conv --iters 20 --w2d-debug $dbg > $O/mb_dbg$dbg.json 2>/dev/null; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace -d $O/pmc_sq -o mb -- python $R/tools/microbench.py conv --iters 3 > /dev/null 2> $O/pmc_sq.err
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS --kernel-trace -d $O/pmc_lds -o mb -- python $R/tools/microbench.py conv --iters 3 > /dev/null 2> $O/pmc_lds.err
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace -d $O/pmc_inst -o mb -- python $R/tools/microbench.py conv --iters 3 > /dev/null 2> $O/pmc_inst.err
cd $R
for d in pmc_sq pmc_lds pmc_inst; do python tools/rocpd_pmc.py $(find $O/$d -name "*.db" | head -1) modconv > $O/$d.md 2>&1; done
find $O -name "*.db" -delete
python - <<'PY'
import json
for d in range(5):
    r=json.load(open(f"gpurun_out/w2d/mb_dbg{d}.json"))
    print(d, {k: round(v["ms"],3) for k,v in r.items() if k.startswith("plain")})
PY
cat $O/pmc_sq.md $O/pmc_lds.md $O/pmc_inst.md
