#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_run7
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -E "Counter_Name" | sed 's/Counter_Name *:\t*//' | sort -u > $O/counter_names.txt; wc -l $O/counter_names.txt
rocm-smi --showclocks --showmemuse --showperflevel 2>/dev/null | head -40 > $O/smi.txt; cat $O/smi.txt
rocm-smi --showmemorypartition --showcomputepartition 2>/dev/null | head -20 >> $O/smi.txt; tail -12 $O/smi.txt
