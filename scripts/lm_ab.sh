#!/bin/bash
# usage: scripts/lm_ab.sh <tag>  (GPU box, repo root): wall time of the 64- and 256-window batches + per-kernel averages of the throughput
# kernels (rocprofv3 kernel trace) and the timeline of the last LM steps
TAG=$1
OUT=$GRAFT_REPO_ROOT/gpurun_out/lm_$TAG
mkdir -p $OUT
python scripts/batched_time.py 64 20 | tee $OUT/time.txt
python scripts/batched_time.py 256 8 | tee -a $OUT/time.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $GRAFT_REPO_ROOT/scripts/batched_run.py 64 > $OUT/trace.log 2>&1
cd $GRAFT_REPO_ROOT
python scripts/kstats.py $OUT/trace 7
python scripts/ktimeline.py $OUT/trace 12
rm -rf $OUT/trace
