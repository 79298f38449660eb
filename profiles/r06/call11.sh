#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_call11
mkdir -p $O
cd $R
AB="python tools/ab_bench.py --reps 3 --steps 30"
for b in 128 256; do
echo "undistort batch $b" | tee -a $O/ab.log
timeout 900 $AB --workload undistort_b64 --bench-args "--placements 2 --single-layout --batch $b" base: ro4:BEVW_UNIT_ROW_ORDER=4 nb8:BEVW_PLAN_NB=8 nb8_ro4:BEVW_PLAN_NB=8,BEVW_UNIT_ROW_ORDER=4 nb4_ro4:BEVW_PLAN_NB=4,BEVW_UNIT_ROW_ORDER=4 2>&1 | tee -a $O/ab.log
done
for b in 16 32 64; do
echo "undistort batch $b" | tee -a $O/ab.log
timeout 900 $AB --workload undistort_b64 --bench-args "--placements 2 --single-layout --batch $b" ro4:BEVW_UNIT_ROW_ORDER=4 nb4_ro4:BEVW_PLAN_NB=4,BEVW_UNIT_ROW_ORDER=4 nb2_ro4:BEVW_PLAN_NB=2,BEVW_UNIT_ROW_ORDER=4 nb16_ro4:BEVW_PLAN_NB=16,BEVW_UNIT_ROW_ORDER=4 2>&1 | tee -a $O/ab.log
done
