#!/bin/bash
# round 4: the (4 quads, 4 rounds) unit class on / off, every stitch workload, interleaved on one box; GPU parity of the stitch tests first
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_run6
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_analytic.py tests/test_camera_shard.py -m gpu -x -q > $O/pytest_stitch.log 2>&1; grep -n "passed\|failed" $O/pytest_stitch.log | tail -1
bash tools/r04/ab.sh bigclass "direct_stitch_b256 blend_b256 blend_balance_b256 undistort_b64 blend_4k direct_stitch_analytic_f32_b64" 2 20 "--placements 2 --single-layout" big1: big0:BEVW_UNIT_BIG=0
bash tools/r04/ab.sh bigclass_dense "direct_stitch_b256" 2 20 "--placements 2 --single-layout --output-pitch dense" big1: big0:BEVW_UNIT_BIG=0
