#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_call9
mkdir -p $O
cd $R
AB="python tools/ab_bench.py --reps 3 --steps 30"
timeout 900 $AB --workload undistort_b64 --bench-args "--placements 2 --single-layout" base: ro1:BEVW_UNIT_ROW_ORDER=1 ro2:BEVW_UNIT_ROW_ORDER=2 ro3:BEVW_UNIT_ROW_ORDER=3 ro4:BEVW_UNIT_ROW_ORDER=4 ro5:BEVW_UNIT_ROW_ORDER=5 ro6:BEVW_UNIT_ROW_ORDER=6 nb4_ro4:BEVW_UNIT_ROW_ORDER=4,BEVW_PLAN_NB=4 2>&1 | tee -a $O/ab.log
timeout 900 $AB --workload blend_4k --bench-args "--placements 2 --single-layout" base: ro4:BEVW_UNIT_ROW_ORDER=4 ro5:BEVW_UNIT_ROW_ORDER=5 ro6:BEVW_UNIT_ROW_ORDER=6 2>&1 | tee -a $O/ab.log
timeout 900 $AB --workload direct_stitch_b256 --bench-args "--placements 2 --single-layout" base: ro4:BEVW_UNIT_ROW_ORDER=4 ro5:BEVW_UNIT_ROW_ORDER=5 ro6:BEVW_UNIT_ROW_ORDER=6 2>&1 | tee -a $O/ab.log
