#!/bin/bash
# usage: scripts/prof_batched.sh <tag>   (run on the GPU box, from the repo root)
# The batched regime (64 windows x 8 000 landmarks: the throughput kernels of lm_kernels.h): rocprofv3 kernel trace, HBM traffic
# (FETCH_SIZE / WRITE_SIZE in separate passes, FETCH_SIZE x 2 on gfx950) and the issue counters VERDICT r01 item 4 asks for
# (VALU / MFMA / LDS busy), each aggregated per kernel; the per-launch raw tables are dropped.
TAG=$1
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_batched_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/scripts/batched_run.py 64"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o p -- $CMD > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o p -- $CMD > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/pmc_issue -o p -- $CMD > $OUT/pmc_issue.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc_issue2 -o p -- $CMD > $OUT/pmc_issue2.log 2>&1
cd $GRAFT_REPO_ROOT
python scripts/summarize_prof.py $OUT > $OUT/summary.json
python scripts/pmc_agg.py $OUT > $OUT/counters_per_kernel.txt
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
rm -rf $OUT/trace $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_issue $OUT/pmc_issue2
cat $OUT/summary.json
