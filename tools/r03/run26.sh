#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_run26
mkdir -p $O
cd $R
timeout 1500 python tools/ab_bench.py --workload direct_stitch_b256 --reps 2 --steps 20 \
  base: nb16:BEVW_PLAN_NB=16 nb12:BEVW_PLAN_NB=12 gm:BEVW_PLAN_GROUPMAJOR=1 gm_nb16:BEVW_PLAN_GROUPMAJOR=1,BEVW_PLAN_NB=16 g768:BEVW_UNIT_GROUPS=768 2>&1 | tee $O/ab.log
