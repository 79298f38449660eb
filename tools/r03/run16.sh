#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_run16
mkdir -p $O
cd $R
timeout 900 python tools/r03/placement.py --mode vmm --trials 3 --steps 10 > $O/vmm.log 2>&1; cat $O/vmm.log
timeout 300 python tools/r03/placement.py --mode realloc --trials 4 --steps 10 > $O/realloc.log 2>&1; cat $O/realloc.log
