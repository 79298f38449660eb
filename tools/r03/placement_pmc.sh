#!/bin/bash
# Counters of the stitch kernel per PLACEMENT: tools/r03/placement.py (one process, buffers re-allocated between trials) under rocprofv3,
# one pass per counter set; tools/r03/placement_table.py joins per trial: kernel duration, memory latency (LEVEL / requests), credit stalls.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r03_placement_pmc}
MODE=${2:-realloc}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
while read -r set; do
  [ -z "$set" ] && continue
  i=$((i+1)); rm -rf /tmp/pp_$i
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pp_$i -- python $R/tools/r03/placement.py --mode $MODE --trials 10 --steps 6 > $O/pass_$i.log 2>&1
  f=$(find /tmp/pp_$i -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f $O/pass_${i}_counters.csv
  f=$(find /tmp/pp_$i -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && cp $f $O/pass_${i}_trace.csv
  python $R/tools/r03/placement_table.py $O/pass_${i}_counters.csv $O/pass_${i}_trace.csv 9 | tee $O/pass_${i}_table.txt
done <<'SETS'
TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_sum
TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_GMI_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_GMI_CREDIT_STALL_sum
TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum
TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum
SETS
