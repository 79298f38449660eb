#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_run28
mkdir -p $O
cd $R
for w in direct_stitch_b256 blend_b256; do
timeout 900 python tools/ab_bench.py --workload $w --reps 2 --steps 20 w3: w4:BEVW_LIB_PATH=$R/build_var/libbevwarp_w4.so 2>&1 | tee -a $O/ab.log
done
