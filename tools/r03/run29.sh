#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_run29
mkdir -p $O
cd $R
timeout 900 python tools/ab_bench.py --workload blend_4k --reps 2 --steps 20 nb4: nb8:BEVW_PLAN_NB=8 nb16:BEVW_PLAN_NB=16 nb32:BEVW_PLAN_NB=32 2>&1 | tee $O/ab.log
timeout 900 python tools/ab_bench.py --workload undistort_b64 --reps 2 --steps 20 nb8: nb16:BEVW_PLAN_NB=16 nb32:BEVW_PLAN_NB=32 2>&1 | tee -a $O/ab.log
