#!/bin/bash
# One gpurun call = one A/B block on one box (tools/ab_bench.py interleaves the variants).
#   gpurun --timeout 1200 -- 'bash tools/ab.sh NAME "direct_stitch_b256 blend_b256" 2 20 "--placements 2 --single-layout" nb8: nb16:BEVW_PLAN_NB=16'
# NAME labels gpurun_out/ab_NAME/ab.log; then workloads (quoted list), reps, steps, extra bench.py arguments (quoted), label:ENV=..,ENV=.. variants.
R=${GRAFT_REPO_ROOT:-$(pwd)}
N=$1; W=$2; REPS=$3; STEPS=$4; BA=$5; shift 5
O=$R/gpurun_out/ab_$N
mkdir -p $O
cd $R
for w in $W; do
  timeout 900 python tools/ab_bench.py --workload $w --reps $REPS --steps $STEPS --bench-args "$BA" "$@" 2>&1 | tee -a $O/ab.log
done
