#!/bin/bash
# A/B of partition / chunk parameters of the unit schedule (env knobs only)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_run2
mkdir -p $O
cd $R
timeout 900 python tools/ab_bench.py --workload direct_stitch_b256 --reps 4 --steps 30 \
  base: nb16:BEVW_PLAN_NB=16 nb4:BEVW_PLAN_NB=4 lc1:BEVW_UNIT_LINE_COST=1 lc3:BEVW_UNIT_LINE_COST=3 \
  root128x64:BEVW_UNIT_ROOT_W=128 root256x128:BEVW_UNIT_ROOT_H=128 root256x32:BEVW_UNIT_ROOT_H=32 g768:BEVW_UNIT_GROUPS=768 > $O/ab.log 2>&1
cat $O/ab.log
