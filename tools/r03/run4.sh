#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_run4
mkdir -p $O
cd $R
timeout 300 python tools/r03/placement.py --mode arena --trials 10 > $O/placement_arena.log 2>&1; cat $O/placement_arena.log
timeout 300 python tools/r03/placement.py --mode joint --trials 8 > $O/placement_joint.log 2>&1; cat $O/placement_joint.log
BEVW_PLAN_UNITS=0 timeout 300 python tools/r03/placement.py --mode realloc --trials 8 > $O/placement_realloc_r02sched.log 2>&1; cat $O/placement_realloc_r02sched.log
cd /tmp && export TMPDIR=/tmp
for i in 1 2 3 4 5 6; do
  rm -rf /tmp/pc_$i
  timeout 120 rocprofv3 --pmc TCC_EA0_RDREQ TCC_EA0_WRREQ TCC_REQ --kernel-trace --output-format json -d /tmp/pc_$i -- python $R/bench.py --workload direct_stitch_b256 --steps 4 --warmup 1 --no-cpu-baseline > /tmp/pc_$i.log 2>&1
  f=$(find /tmp/pc_$i -name "*results.json" | head -1)
  ls -la $f
  [ -n "$f" ] && python $R/tools/r03/channel_table.py $f > $O/channels_$i.txt 2>&1; tail -25 $O/channels_$i.txt
done
