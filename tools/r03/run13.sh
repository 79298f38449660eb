#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_run13
mkdir -p $O
cd $R
for fs in 0 1; do
  echo "== fstride $fs"
  BEVW_PLAN_FSTRIDE=$fs timeout 600 python tools/r03/placement.py --mode cycle --trials 4 --steps 10 > $O/cycle_fs$fs.log 2>&1; grep "alone" $O/cycle_fs$fs.log
  BEVW_PLAN_FSTRIDE=$fs timeout 600 python tools/r03/placement.py --mode flags --trials 2 --steps 10 > $O/flags_fs$fs.log 2>&1; cat $O/flags_fs$fs.log
done
BEVW_PLAN_FSTRIDE=1 timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "batch_256 or repo_data" 2>&1 | tail -3
