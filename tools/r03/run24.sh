#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_run24
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; grep -E "passed|failed|Error" $O/pytest_gpu.log | tail -3
for w in direct_stitch_b256 blend_b256 blend_4k blend_balance_b256; do
timeout 900 python tools/ab_bench.py --workload $w --reps 2 --steps 20 nodouble:BEVW_UNIT_OWN_DOUBLE=0 now: 2>&1 | tee -a $O/ab.log
done
