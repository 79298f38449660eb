#!/bin/bash
# Partial-sector write counters of the stitch kernel per env variant (profiles/r03/write_counters_spatial_skew.log).
#   gpurun --timeout 600 -- 'bash tools/r03/write_counters.sh base: skew64:BEVW_UNIT_SKEW=64'
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_write_counters
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  L=${v%%:*}; E=$(echo "${v#*:}" | tr ',' ' ')
  rm -rf /tmp/pw_$L
  env $E timeout 120 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum --output-format csv -d /tmp/pw_$L -- \
    python $R/bench.py --workload direct_stitch_b256 --steps 3 --warmup 1 --placements 1 --single-layout --no-cpu-baseline > /tmp/pw_$L.log 2>&1
  f=$(find /tmp/pw_$L -name "*counter_collection.csv" | head -1)
  python - $f $L <<'PY'
import csv,sys
from collections import defaultdict
t=defaultdict(float);n=defaultdict(int)
for r in csv.DictReader(open(sys.argv[1])):
    if "k_plan_all" in r["Kernel_Name"]:
        t[r["Counter_Name"]]+=float(r["Counter_Value"]);n[r["Counter_Name"]]+=1
print(sys.argv[2], {k:round(t[k]/n[k]) for k in t})
PY
done 2>&1 | tee $O/write_counters.log
