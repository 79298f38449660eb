#!/bin/bash
# rocprofv3 kernel stats of one workload: gpurun -- 'bash tools/r03/kstats.sh tag workload "ENV=.." [bench args]'
TAG=${1:-kstats}; W=${2:-direct_stitch_b256}; E=${3:-}; shift 3
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ks
env $E timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python $R/bench.py --workload $W --steps 10 --warmup 2 --placements 1 --no-cpu-baseline "$@" > /tmp/ks.log 2>&1
f=$(find /tmp/ks -name "*kernel_stats.csv" | head -1)
cp $f $O/kernel_stats_$W.csv
head -14 $f | cut -d, -f1-5 | cut -c1-150
