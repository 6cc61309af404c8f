#!/bin/bash
# usage: scripts/pmc_quick.sh [windows]  (GPU box, repo root): issue counters of the batched run per kernel, two passes
W=${1:-64}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pq_$$
mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/scripts/batched_run.py $W"
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/a -o p -- $CMD > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/b -o p -- $CMD > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; python scripts/pmc_agg.py $OUT | grep -v "fillBuffer\|k_reset\|scatter\|k_decide"; rm -rf $OUT
