#!/bin/bash
# One gpurun call = one A/B block on one box (tools/ab_bench.py interleaves the variants).
#   gpurun --timeout 1200 -- 'bash tools/r03/ab.sh NAME "direct_stitch_b256 blend_b256" 2 20 nb8: nb16:BEVW_PLAN_NB=16'
# NAME labels gpurun_out/r03_ab_NAME/ab.log; then workloads (quoted list), reps, steps, and label:ENV=..,ENV=.. variants.
# Every sweep of profiles/r03/sweeps.log was one such call (the "(runN)" marks there are the order they were made in).
R=${GRAFT_REPO_ROOT:-$(pwd)}
N=$1; W=$2; REPS=$3; STEPS=$4; shift 4
O=$R/gpurun_out/r03_ab_$N
mkdir -p $O
cd $R
: > $O/ab.log
for w in $W; do
  timeout 900 python tools/ab_bench.py --workload $w --reps $REPS --steps $STEPS "$@" 2>&1 | tee -a $O/ab.log
done
