#!/bin/bash
# is the placement spread an address-translation effect?  UTCL1 (per-CU TLB) counters per k_plan_all dispatch while placement_probe.py
# re-allocates the buffers (15 dispatches per trial: 3 warm-up + 12 timed); the probe prints the per-trial step time
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02_run43
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for set in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum" "TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_PERMISSION_MISS_sum"; do
  rm -rf /tmp/pq
  timeout 300 rocprofv3 --pmc $set --output-format csv -d /tmp/pq -- python $R/tools/r02/placement_probe.py > /tmp/probe.log 2>&1
  grep trial /tmp/probe.log > $O/probe_times.txt
  python - $(find /tmp/pq -name "*counter_collection.csv" | head -1) $O/probe_times.txt <<'PY'
import csv, sys
from collections import defaultdict
rows = defaultdict(dict)
for r in csv.DictReader(open(sys.argv[1])):
    if "k_plan_all" in r["Kernel_Name"]:
        rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(rows)
times = [float(l.split("median")[1].split()[0]) for l in open(sys.argv[2])]
print("dispatches", len(ids), "trials", len(times))
per = len(ids) // max(1, len(times))
for t, ms in enumerate(times):
    chunk = ids[t * per:(t + 1) * per]
    agg = defaultdict(float)
    for i in chunk:
        for k, v in rows[i].items(): agg[k] += v / len(chunk)
    print("trial %d  %.4f ms  " % (t, ms) + "  ".join("%s %.3g" % (k.replace("TCP_UTCL1_", "").replace("_sum", ""), v) for k, v in sorted(agg.items())))
PY
done
