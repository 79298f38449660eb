#!/bin/bash
# gpurun -- 'bash tools/r03/soak.sh FIRST LAST [TIMEOUT_S] [analytic]': tools/soak_stitch.py (table modes) or tools/soak_analytic.py on the GPU box
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_soak
mkdir -p $O
cd $R
T=soak_stitch; [ "${4:-}" = "analytic" ] && T=soak_analytic
timeout ${3:-1500} python tools/$T.py ${1:-0} ${2:-1500} > $O/${T}_${1:-0}_${2:-1500}.log 2>&1; tail -5 $O/${T}_${1:-0}_${2:-1500}.log
