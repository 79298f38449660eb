#!/bin/bash
# round 4: JPEG + tools tests (EXIF orientations), prefetch depth 2 / 4 of the dense unit classes on every stitch workload
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_run7
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_jpeg_gpu.py tests/test_tools.py tests/test_jpeg_goldens.py -m gpu -x -q > $O/pytest_jpeg.log 2>&1; grep -n "passed\|failed" $O/pytest_jpeg.log | tail -1
BEVW_LIB_PATH=$R/build_var/libbevwarp_depth4.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "repo_data or bench_configuration or full_size or balance" > $O/pytest_depth4.log 2>&1; grep -n "passed\|failed" $O/pytest_depth4.log | tail -1
bash tools/r04/ab.sh depth "direct_stitch_b256 blend_b256 blend_balance_b256 undistort_b64 blend_4k" 2 20 "--placements 2 --single-layout" depth2: depth4:BEVW_LIB_PATH=$R/build_var/libbevwarp_depth4.so
bash tools/r04/ab.sh depth_dense "direct_stitch_b256" 2 20 "--placements 2 --single-layout --output-pitch dense" depth2: depth4:BEVW_LIB_PATH=$R/build_var/libbevwarp_depth4.so
