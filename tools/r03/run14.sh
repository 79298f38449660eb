#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_run14
mkdir -p $O
cd $R
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
timeout 900 python tools/ab_bench.py --workload direct_stitch_b256 --reps 3 --steps 20 \
  v1:BEVW_LIB_PATH=$R/build_var/libbevwarp_v1.so v2: v2_noalign:BEVW_UNIT_ALIGN_LINES=0 v2_noempty:BEVW_UNIT_OWN_EMPTY=0 v2_lc1:BEVW_UNIT_LINE_COST=1 > $O/ab_direct.log 2>&1; cat $O/ab_direct.log
bash tools/r03/pmc_class.sh direct_stitch_b256 r03_pmc2 0 > $O/pmc.log 2>&1; tail -32 $O/pmc.log
