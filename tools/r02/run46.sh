#!/bin/bash
# k_lum_groups: packed-fp32 HSV -> BGR, 24-bit multiplies, buffer addressing -- GPU suite, kernel time old / new, config 4 A/B
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_lum.log 2>&1; grep -n "passed\|failed\|rror" gpurun_out/pytest_lum.log | tail -3
cd /tmp && export TMPDIR=/tmp
for v in head new head new; do
  rm -rf /tmp/kt
  if [ $v = head ]; then export BEVW_LIB_PATH=$R/build_var/libbevwarp_head.so; else unset BEVW_LIB_PATH; fi
  timeout 180 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $R/bench.py --workload blend_balance_b256 --steps 10 --warmup 2 --no-cpu-baseline > /tmp/kt.log 2>&1
  python - $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $v <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "k_lum_groups" in r["Name"]:
        print("k_lum_groups %s avg %8.1f us" % (sys.argv[2], float(r["AverageNs"]) / 1e3))
PY
done
unset BEVW_LIB_PATH
cd $R
python tools/ab_bench.py --workload blend_balance_b256 --reps 6 head:BEVW_LIB_PATH=build_var/libbevwarp_head.so new:X=1
