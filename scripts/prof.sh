#!/bin/bash
# usage: scripts/prof.sh <tag> [windows]   (run on the GPU box, from the repo root)
TAG=$1; NW=${2:-1}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $GRAFT_REPO_ROOT/scripts/gpu_time.py $NW > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $OUT/pmc1 -o p -- python $GRAFT_REPO_ROOT/scripts/gpu_time.py $NW > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc2 -o p -- python $GRAFT_REPO_ROOT/scripts/gpu_time.py $NW > $OUT/pmc2.log 2>&1
find $OUT -name "*.csv" | head -20
