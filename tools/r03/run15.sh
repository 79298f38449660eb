#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_run15
mkdir -p $O
cd $R
timeout 1200 python tools/ab_bench.py --workload direct_stitch_b256 --reps 3 --steps 20 \
  base: nb16:BEVW_PLAN_NB=16 gm:BEVW_PLAN_GROUPMAJOR=1 gm_nb16:BEVW_PLAN_GROUPMAJOR=1,BEVW_PLAN_NB=16 nb32:BEVW_PLAN_NB=32 > $O/ab_direct.log 2>&1; cat $O/ab_direct.log
BEVW_PLAN_GROUPMAJOR=1 timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "batch_256 or repo_data" 2>&1 | tail -2
