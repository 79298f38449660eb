#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_run6
mkdir -p $O
cd $R
timeout 600 python tools/r03/placement.py --mode delta --trials 40 --steps 10 > $O/placement_delta.log 2>&1; cat $O/placement_delta.log
timeout 600 python tools/r03/placement.py --mode realloc --trials 12 --steps 10 > $O/placement_realloc.log 2>&1; cat $O/placement_realloc.log
