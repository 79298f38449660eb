#!/bin/bash
# sector-aligned re-cut of block-tile runs: request counters and kernel time of k_plan_block, straight / re-cut
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for sh in 0 1; do
  rm -rf /tmp/pq /tmp/kt
  BEVW_PLAN_SHEAR=$sh timeout 90 rocprofv3 --pmc TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum TCC_EA0_WRREQ_sum --output-format csv -d /tmp/pq -- python $R/bench.py --workload direct_stitch_b256 --steps 3 --warmup 1 --no-cpu-baseline > /tmp/pq.log 2>&1
  BEVW_PLAN_SHEAR=$sh timeout 180 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $R/bench.py --workload direct_stitch_b256 --steps 10 --warmup 2 --no-cpu-baseline > /tmp/kt.log 2>&1
  python - $(find /tmp/pq -name "*counter_collection.csv" | head -1) $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $sh <<'PY'
import csv, sys
from collections import defaultdict
tot = defaultdict(float); n = defaultdict(int)
for r in csv.DictReader(open(sys.argv[1])):
    k = r.get("Kernel_Name", "")
    if "k_plan_block" in k or "k_plan_all" in k:
        key = (k.split("(")[0][-40:], r["Counter_Name"])
        tot[key] += float(r["Counter_Value"]); n[key] += 1
for key in sorted(tot): print("shear", sys.argv[3], key[0], key[1], "%.0f" % (tot[key] / n[key]))
for r in csv.DictReader(open(sys.argv[2])):
    if "k_plan_block" in r["Name"] or "k_plan_all" in r["Name"]:
        print("shear", sys.argv[3], r["Name"].split("(")[0][-40:], "avg %.1f us" % (float(r["AverageNs"]) / 1e3))
PY
done
