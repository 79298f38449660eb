#!/bin/bash
# k_vsum launch shape: blocks per frame and contiguous / interleaved block ranges (kernel-trace averages)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for v in "8 0" "8 1" "16 0" "16 1" "32 1" "4 1" "64 1" "8 0"; do
  set -- $v
  rm -rf /tmp/kt
  BEVW_VSUM_BPF=$1 BEVW_VSUM_MODE=$2 timeout 180 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $R/bench.py --workload blend_balance_b256 --steps 10 --warmup 2 --no-cpu-baseline > /tmp/kt.log 2>&1
  python - $(find /tmp/kt -name "*kernel_stats.csv" | head -1) "$1 $2" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "k_vsum" in r["Name"]:
        print("k_vsum bpf/mode %s avg %8.1f us" % (sys.argv[2], float(r["AverageNs"]) / 1e3))
PY
done
