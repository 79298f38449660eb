#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_run12
mkdir -p $O
cd $R
timeout 600 python tools/r03/placement.py --mode cycle --trials 6 --steps 10 > $O/cycle.log 2>&1; cat $O/cycle.log
