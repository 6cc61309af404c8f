#!/bin/bash
# usage: scripts/prof_round.sh <tag>  (GPU box, repo root): the round's committed profiles in one call — bench line + config-2 kernel
# trace + HBM PMC passes (prof_bench.sh), then the 64-window batch: trace, HBM PMC passes, issue counters (prof_batched.sh)
TAG=$1
bash scripts/prof_bench.sh $TAG > /dev/null 2>&1
bash scripts/prof_batched.sh $TAG > /dev/null 2>&1
ls gpurun_out/prof_$TAG gpurun_out/prof_batched_$TAG
tail -c 400 gpurun_out/prof_$TAG/bench.err
