#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_run10
mkdir -p $O
cd $R
timeout 600 python tools/r03/placement.py --mode flags --trials 6 --steps 10 > $O/placement_flags.log 2>&1; cat $O/placement_flags.log
