#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_call10
mkdir -p $O
cd $R
AB="python tools/ab_bench.py --reps 3 --steps 30"
for b in 16 32 64 128 256; do
echo "undistort batch $b" | tee -a $O/ab.log
timeout 900 $AB --workload undistort_b64 --bench-args "--placements 2 --single-layout --batch $b" base: ro4:BEVW_UNIT_ROW_ORDER=4 2>&1 | tee -a $O/ab.log
done
for b in 32 64 128; do
echo "direct stitch batch $b" | tee -a $O/ab.log
timeout 900 $AB --workload direct_stitch_b256 --bench-args "--placements 2 --single-layout --batch $b" base: ro4:BEVW_UNIT_ROW_ORDER=4 2>&1 | tee -a $O/ab.log
done
echo "blend_4k batch 64" | tee -a $O/ab.log
timeout 900 $AB --workload blend_4k --bench-args "--placements 2 --single-layout --batch 64" base: ro4:BEVW_UNIT_ROW_ORDER=4 2>&1 | tee -a $O/ab.log
