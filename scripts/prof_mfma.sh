#!/bin/bash
# usage: scripts/prof_mfma.sh <tag>   (GPU box, repo root): FP64 matrix-core counters of the bench workload and of a
# config-4 solve (HBM-resident reduced system): MFMA instruction counts, MFMA busy cycles, SQ busy cycles.
TAG=$1
OUT=$GRAFT_REPO_ROOT/gpurun_out/mfma_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/c2 -o p -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --batch 0 --steps 20 > $OUT/c2.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/c4 -o p -- python $GRAFT_REPO_ROOT/scripts/gpu_time_big.py c4 > $OUT/c4.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<PY
import csv, glob, collections, json
out = {}
for cfg in ("c2", "c4"):
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % cfg, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("sadvio::", "")
            if k.startswith("__amd"): continue
            a = acc[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    out[cfg] = {k: {c: round(v[0] / max(v[1], 1), 1) for c, v in d.items()} for k, d in acc.items()}
    for k, d in out[cfg].items():
        if d.get("SQ_BUSY_CYCLES"): d["mfma_busy_over_sq_busy"] = round(d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / d["SQ_BUSY_CYCLES"], 4)
print(json.dumps(out, indent=1))
PY
