#!/bin/bash
# usage: scripts/prof_mfma.sh <tag>   (GPU box, repo root): FP64 matrix-core counters (MFMA instruction counts, MFMA busy cycles, SQ busy
# cycles) of the kernels that carry the path's dense contractions TODAY: the single-window bench workload (k_build, k_solve<0>: chol16),
# the config-3 shaped window with the dense prior (k_wchol_*: the wide-panel Cholesky with the look-ahead block), the config-4 window
# (banded solver) and the marginalisation loop (k_mgemm Schur complement, k_tri_level triangular inverse, k_jacobi_mma eigen form).
TAG=$1
OUT=$GRAFT_REPO_ROOT/gpurun_out/mfma_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PMC="SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE"
rocprofv3 --pmc $PMC --output-format csv -d $OUT/c2 -o p -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-vio --no-marginalize --batch 0 --steps 5 --warmup 1 --solves-per-step 20 > $OUT/c2.log 2>&1
rocprofv3 --pmc $PMC --output-format csv -d $OUT/c3dense -o p -- python $GRAFT_REPO_ROOT/scripts/gpu_time_c3.py dense > $OUT/c3dense.log 2>&1
rocprofv3 --pmc $PMC --output-format csv -d $OUT/c4 -o p -- python $GRAFT_REPO_ROOT/scripts/gpu_time_big.py c4 > $OUT/c4.log 2>&1
rocprofv3 --pmc $PMC --output-format csv -d $OUT/marg -o p -- python $GRAFT_REPO_ROOT/scripts/gpu_time_backend_step.py 300 1 > $OUT/marg.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<PY
import csv, glob, collections, json
out = {}
for cfg in ("c2", "c3dense", "c4", "marg"):
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % cfg, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("sadvio::", "")
            if k.startswith("__amd"): continue
            a = acc[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    out[cfg] = {k: {c: round(v[0] / max(v[1], 1), 1) for c, v in d.items()} for k, d in acc.items()}
    for k, d in out[cfg].items():
        if d.get("SQ_BUSY_CYCLES"): d["mfma_busy_over_sq_busy"] = round(d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / d["SQ_BUSY_CYCLES"], 4)
        if d.get("GRBM_GUI_ACTIVE"): d["mfma_busy_cycles_per_simd_over_kernel_cycles"] = round(d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024.0 / (d["GRBM_GUI_ACTIVE"] / 8.0), 4)
json.dump(out, open("$OUT/mfma_counters.json", "w"), indent=1)
print(json.dumps({c: {k: v for k, v in d.items() if v.get("SQ_INSTS_VALU_MFMA_F64", 0) > 0} for c, d in out.items()}, indent=1)[:6000])
PY
rm -rf $OUT/c2 $OUT/c3dense $OUT/c4 $OUT/marg
