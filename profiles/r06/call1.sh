#!/bin/bash
# GPU call 1 of round 6: store-format microbenchmark + parity of the new default build + format / blend A/B blocks
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_call1
mkdir -p $O
cd $R
timeout 300 ./tools/store_pattern format > $O/store_format.log 2>&1; cat $O/store_format.log
timeout 1200 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; tail -5 $O/pytest_gpu.log
V=build_var
AB="python tools/ab_bench.py --reps 3 --steps 20"
timeout 900 $AB --workload direct_stitch_b256 --bench-args "--placements 2 --single-layout" r05:BEVW_LIB_PATH=$V/libbevwarp_r05.so s0:BEVW_LIB_PATH=$V/libbevwarp_s0.so s1: \
   mem_s0:BEVW_LIB_PATH=$V/libbevwarp_x1s0.so mem_s1:BEVW_LIB_PATH=$V/libbevwarp_x1s1.so st_s0:BEVW_LIB_PATH=$V/libbevwarp_x3s0.so st_s1:BEVW_LIB_PATH=$V/libbevwarp_x3s1.so 2>&1 | tee -a $O/ab.log
timeout 600 $AB --workload direct_stitch_b256 --bench-args "--placements 2 --single-layout --output-pitch dense" r05:BEVW_LIB_PATH=$V/libbevwarp_r05.so s1: s2:BEVW_LIB_PATH=$V/libbevwarp_s2.so 2>&1 | tee -a $O/ab.log
timeout 600 $AB --workload blend_b256 --bench-args "--placements 2 --single-layout" r05:BEVW_LIB_PATH=$V/libbevwarp_r05.so s0:BEVW_LIB_PATH=$V/libbevwarp_s0.so s1: 2>&1 | tee -a $O/ab.log
timeout 600 $AB --workload blend_balance_b256 --bench-args "--placements 2 --single-layout" r05:BEVW_LIB_PATH=$V/libbevwarp_r05.so s0:BEVW_LIB_PATH=$V/libbevwarp_s0.so s1: 2>&1 | tee -a $O/ab.log
timeout 600 $AB --workload undistort_b64 --bench-args "--placements 2 --single-layout" r05:BEVW_LIB_PATH=$V/libbevwarp_r05.so s0:BEVW_LIB_PATH=$V/libbevwarp_s0.so s1: 2>&1 | tee -a $O/ab.log
timeout 600 $AB --workload blend_4k --bench-args "--placements 2 --single-layout" r05:BEVW_LIB_PATH=$V/libbevwarp_r05.so s0:BEVW_LIB_PATH=$V/libbevwarp_s0.so s1: 2>&1 | tee -a $O/ab.log
