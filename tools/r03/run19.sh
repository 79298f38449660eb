#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_run19
mkdir -p $O
cd $R
for w in undistort_b64 blend_b256 blend_balance_b256 blend_4k; do
timeout 900 python tools/ab_bench.py --workload $w --reps 2 --steps 20 v1:BEVW_LIB_PATH=$R/build_var/libbevwarp_v1.so r02sched:BEVW_PLAN_UNITS=0 now: classorder:BEVW_PLAN_SPATIAL=0 2>&1 | tee -a $O/ab.log
done
