#!/bin/bash
# usage: scripts/lm_ab.sh <tag>  (GPU box, repo root): wall time of the 64-window batch + per-kernel averages (rocprofv3 kernel trace),
# with k_diag beside the tile kernels (default) and serialised behind them (SADVIO_NO_PAR: every kernel's stand-alone duration)
TAG=$1
OUT=$GRAFT_REPO_ROOT/gpurun_out/lm_$TAG
mkdir -p $OUT
python scripts/batched_time.py 64 20 | tee $OUT/time.txt
python scripts/batched_time.py 256 8 | tee -a $OUT/time.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $GRAFT_REPO_ROOT/scripts/batched_run.py 64 > $OUT/trace.log 2>&1
SADVIO_NO_PAR=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_nopar -o t -- python $GRAFT_REPO_ROOT/scripts/batched_run.py 64 > $OUT/trace_nopar.log 2>&1
cd $GRAFT_REPO_ROOT
echo "--- parallel k_diag"; python scripts/kstats.py $OUT/trace 7
echo "--- serial (stand-alone durations)"; python scripts/kstats.py $OUT/trace_nopar 7
rm -rf $OUT/trace $OUT/trace_nopar
