#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_run3
mkdir -p $O
cd $R
timeout 300 python tools/r03/placement.py --mode realloc --trials 10 > $O/placement_realloc.log 2>&1; cat $O/placement_realloc.log
timeout 300 python tools/r03/placement.py --mode offset --trials 10 > $O/placement_offset.log 2>&1; cat $O/placement_offset.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -i -E "TCC_EA0_(RD|WR)REQ|TCC_EA0_RDREQ_DRAM|TCC_REQ\b|dimension|DIMENSION" | head -40 > $O/counters.txt; head -40 $O/counters.txt
rocprofv3 -L 2>/dev/null | grep -i -B2 -A12 "TCC_EA0_RDREQ$\|Name.*TCC_EA0_RDREQ\b" | head -60 >> $O/counters.txt; tail -40 $O/counters.txt
