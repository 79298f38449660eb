#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_soak
mkdir -p $O
cd $R
timeout ${3:-1500} python tools/soak_stitch.py ${1:-0} ${2:-1500} > $O/soak_${1:-0}_${2:-1500}.log 2>&1; tail -5 $O/soak_${1:-0}_${2:-1500}.log
