#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_run11
mkdir -p $O
cd $R
for rot in 0 1; do
  BEVW_PLAN_ROTATE=$rot timeout 600 python tools/r03/placement.py --mode flags --trials 4 --steps 10 > $O/flags_rot$rot.log 2>&1; echo "== rotate $rot"; cat $O/flags_rot$rot.log
done
for nb in 4 16; do
  BEVW_PLAN_NB=$nb timeout 600 python tools/r03/placement.py --mode flags --trials 2 --steps 10 > $O/flags_nb$nb.log 2>&1; echo "== rotate 1 nb $nb"; cat $O/flags_nb$nb.log
done
