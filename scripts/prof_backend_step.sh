#!/bin/bash
# usage: scripts/prof_backend_step.sh <tag>   (GPU box, repo root): rocprofv3 kernel trace of the per-key-frame back-end step
# (scripts/gpu_time_backend_step.py: marginalize -> [sparsify] -> set_windows -> solve -> get_deltas, config-3 shape).
TAG=$1
OUT=$GRAFT_REPO_ROOT/gpurun_out/bstep_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o p -- python $GRAFT_REPO_ROOT/scripts/gpu_time_backend_step.py 300 4 > $OUT/run.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $OUT/trace -name '*kernel_stats.csv' | head -1)
cp "$f" $OUT/kernel_stats.csv
python - <<PY
import csv
rows = list(csv.DictReader(open("$OUT/kernel_stats.csv")))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:45]:
    print(r["Name"].split("(")[0].replace("void ", "").replace("sadvio::", "")[:48].ljust(48), r["Calls"].rjust(7), "%9.1f" % (float(r["AverageNs"]) / 1e3), "%9.2f" % (float(r["TotalDurationNs"]) / 1e6))
PY
rm -rf $OUT/trace
