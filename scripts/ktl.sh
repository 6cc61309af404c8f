#!/bin/bash
# usage: scripts/ktl.sh [n_rows] [windows]  (GPU box, repo root): kernel timeline of the last solve of the batched run (rocprofv3 kernel trace)
N=${1:-12}; W=${2:-64}
OUT=$GRAFT_REPO_ROOT/gpurun_out/tl_$$
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python $GRAFT_REPO_ROOT/scripts/batched_run.py $W > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; python scripts/ktimeline.py $OUT $N; rm -rf $OUT
