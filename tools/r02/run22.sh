#!/bin/bash
# block-staged schedule: per-kernel times (per-class launches) and request counters per kernel, block tiles on / off
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02_run22
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for blk in 0 1; do
  for ol in 0 1; do
    rm -rf /tmp/kt
    BEVW_PLAN_BLOCK=$blk BEVW_PLAN_ONELAUNCH=$ol timeout 180 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $R/bench.py --workload direct_stitch_b256 --steps 10 --warmup 2 --no-cpu-baseline > /tmp/kt.log 2>&1
    cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $O/kernel_stats_block${blk}_onelaunch${ol}.csv
  done
  i=0
  for set in "GRBM_GUI_ACTIVE TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum" "TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS"; do
    i=$((i+1)); rm -rf /tmp/pq
    BEVW_PLAN_BLOCK=$blk BEVW_PLAN_ONELAUNCH=0 timeout 90 rocprofv3 --pmc $set --output-format csv -d /tmp/pq -- python $R/bench.py --workload direct_stitch_b256 --steps 3 --warmup 1 --no-cpu-baseline > /tmp/pq.log 2>&1
    f=$(find /tmp/pq -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && cp $f $O/pmc_block${blk}_pass$i.csv || { echo "pass $i failed"; tail -2 /tmp/pq.log; }
  done
done
cd $R
python - $O <<'PY'
import csv, glob, sys
from collections import defaultdict
O = sys.argv[1]
for blk in (0, 1):
    for ol in (0, 1):
        print("== kernel stats block%d onelaunch%d" % (blk, ol))
        for r in csv.DictReader(open("%s/kernel_stats_block%d_onelaunch%d.csv" % (O, blk, ol))):
            if "k_plan" in r["Name"] and "build" not in r["Name"] and "touch" not in r["Name"]:
                print("   %-70s calls %3s avg %8.1f us" % (r["Name"].split("(")[0][-70:], r["Calls"], float(r["AverageNs"]) / 1e3))
    tot = defaultdict(lambda: defaultdict(float)); n = defaultdict(lambda: defaultdict(int))
    for f in sorted(glob.glob("%s/pmc_block%d_pass*.csv" % (O, blk))):
        for r in csv.DictReader(open(f)):
            k = r.get("Kernel_Name", "")
            if "k_plan" in k and "build" not in k and "touch" not in k:
                k = k.split("(")[0][-50:]
                tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
    print("== counters per launch, block%d" % blk)
    for k in tot:
        print("  ", k)
        for c in tot[k]: print("      %-34s %14.0f" % (c, tot[k][c] / max(1, n[k][c])))
PY
