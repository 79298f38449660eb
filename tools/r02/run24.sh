#!/bin/bash
# compact slices of the sliced class: GPU suite, per-class kernel times and read requests, bench lines
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02_run24
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; grep -n "passed\|failed\|rror" $O/pytest.log | tail -5
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt
BEVW_PLAN_ONELAUNCH=0 timeout 180 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $R/bench.py --workload direct_stitch_b256 --steps 10 --warmup 2 --no-cpu-baseline > /tmp/kt.log 2>&1
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $O/kernel_stats_per_class.csv
rm -rf /tmp/pq
BEVW_PLAN_ONELAUNCH=0 timeout 90 rocprofv3 --pmc TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d /tmp/pq -- python $R/bench.py --workload direct_stitch_b256 --steps 3 --warmup 1 --no-cpu-baseline > /tmp/pq.log 2>&1
cp $(find /tmp/pq -name "*counter_collection.csv" | head -1) $O/pmc_requests.csv
cd $R
python - $O <<'PY'
import csv, sys
from collections import defaultdict
O = sys.argv[1]
for r in csv.DictReader(open(O + "/kernel_stats_per_class.csv")):
    if "k_plan" in r["Name"] and "build" not in r["Name"] and "touch" not in r["Name"]:
        print("   %-70s calls %3s avg %8.1f us" % (r["Name"].split("(")[0][-70:], r["Calls"], float(r["AverageNs"]) / 1e3))
tot = defaultdict(lambda: defaultdict(float)); n = defaultdict(lambda: defaultdict(int))
for r in csv.DictReader(open(O + "/pmc_requests.csv")):
    k = r.get("Kernel_Name", "")
    if "k_plan" in k and "build" not in k and "touch" not in k:
        k = k.split("(")[0][-50:]
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
for k in tot:
    print("  ", k, " ".join("%s %.0f" % (c.replace("TCP_", "").replace("_sum", ""), tot[k][c] / max(1, n[k][c])) for c in tot[k]))
PY
B="python bench.py --steps 40 --warmup 5 --no-cpu-baseline --workload"
res() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('RESULT $1 ms %.4f median %.4f frac %.4f' % (d['roofline']['kernel_ms'], d['roofline'].get('kernel_ms_median', 0), d['roofline']['frac']))"; }
for rep in 1 2; do
for w in direct_stitch_b256 blend_b256 undistort_b64 blend_4k blend_balance_b256; do
  timeout 300 $B $w 2>&1 | tail -1 | res "$w" | tee -a $O/ab.log
done
done
