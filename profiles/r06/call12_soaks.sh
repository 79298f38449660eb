#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_call12
mkdir -p $O
cd $R
BEVW_SOAK_SECONDS=420 timeout 600 python tools/soak_stitch.py 100000 130000 > $O/soak_stitch.log 2>&1; tail -3 $O/soak_stitch.log
BEVW_SOAK_SECONDS=90 timeout 200 python tools/soak_warps.py 100000 130000 > $O/soak_warps.log 2>&1; tail -2 $O/soak_warps.log
BEVW_SOAK_SECONDS=240 timeout 400 python tools/soak_analytic.py 100000 130000 > $O/soak_analytic.log 2>&1; tail -3 $O/soak_analytic.log
BEVW_ANALYTIC_UNITS=0 BEVW_ANALYTIC_FRAMES=1 BEVW_SOAK_SECONDS=120 timeout 300 python tools/soak_analytic.py 130000 160000 > $O/soak_analytic_perpixel.log 2>&1; tail -3 $O/soak_analytic_perpixel.log
BEVW_SOAK_SECONDS=60 timeout 200 python tools/soak_jpeg.py --seed 61 --cases 100000 > $O/soak_jpeg.log 2>&1; tail -2 $O/soak_jpeg.log
