#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_call15
mkdir -p $O
cd $R
BEVW_PLAN_XCDMAP=3 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_analytic.py -m gpu -q -x > $O/pytest_map3.log 2>&1; grep -E "passed|failed" $O/pytest_map3.log
AB="python tools/ab_bench.py --reps 3 --steps 20"
for w in direct_stitch_b256 blend_b256 blend_balance_b256 undistort_b64 blend_4k direct_stitch_analytic_f32_b64; do
timeout 900 $AB --workload $w --bench-args "--placements 2 --single-layout" map1: map3:BEVW_PLAN_XCDMAP=3 map3_nb8:BEVW_PLAN_XCDMAP=3,BEVW_PLAN_NB=8 map3_nb32:BEVW_PLAN_XCDMAP=3,BEVW_PLAN_NB=32 2>&1 | tee -a $O/ab.log
done
timeout 600 $AB --workload direct_stitch_b256 --bench-args "--placements 2 --single-layout --output-pitch dense" map1: map3:BEVW_PLAN_XCDMAP=3 2>&1 | tee -a $O/ab.log
