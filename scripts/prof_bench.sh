#!/bin/bash
# usage: scripts/prof_bench.sh <tag>   (run on the GPU box, from the repo root)
# bench line + rocprofv3 kernel trace of the same command + HBM-traffic PMC passes (FETCH_SIZE and WRITE_SIZE
# in separate passes, MI355X_MICROARCH.md "HBM": they do not fit one pass; FETCH_SIZE x2 on gfx950).
TAG=$1
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
python bench.py > $OUT/bench.json 2> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
# a short run of the same command: 5 x 20 timed solves (~10^4 kernel launches; the default 2000 solves make traces of > 64 MiB)
CMD="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-vio --no-marginalize --batch 0 --steps 5 --warmup 1 --solves-per-step 20"   # the config-2 legs only: one variant per kernel in the trace
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o p -- $CMD > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o p -- $CMD > $OUT/pmc_write.log 2>&1
cd $GRAFT_REPO_ROOT
python scripts/summarize_prof.py $OUT > $OUT/summary.json
# keep the per-kernel tables, drop the per-launch raw traces
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
rm -rf $OUT/trace $OUT/pmc_fetch $OUT/pmc_write
cat $OUT/bench.json; cat $OUT/summary.json
