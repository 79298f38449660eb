#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_run8
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pc
timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ TCC_EA0_WRREQ --kernel-trace --output-format json -d /tmp/pc -- python $R/tools/r03/placement.py --mode realloc --trials 10 --steps 6 > $O/placement.log 2>&1
f=$(find /tmp/pc -name "*results.json" | head -1)
python $R/tools/r03/channel_table.py $f 9 | tee $O/channels.txt
