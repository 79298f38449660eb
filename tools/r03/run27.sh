#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_run27
mkdir -p $O
cd $R
timeout 900 python tools/ab_bench.py --workload direct_stitch_b256 --reps 2 --steps 20 nb8: nb16:BEVW_PLAN_NB=16 nb32:BEVW_PLAN_NB=32 2>&1 | tee $O/ab.log
for w in blend_b256 blend_balance_b256; do
timeout 900 python tools/ab_bench.py --workload $w --reps 2 --steps 20 nb8: nb16:BEVW_PLAN_NB=16 2>&1 | tee -a $O/ab.log
done
