#!/bin/bash
# write-side counters of the strip assignments of tools/store_pattern (6 dispatches per mode; modes 0 1 3 4)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_store
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
$R/tools/store_pattern strips | tee $O/strips.log
for set in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum" "TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_GMI_CREDIT_STALL_sum TCC_EA0_WRREQ_IO_CREDIT_STALL_sum" "TCP_TCC_WRITE_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCC_WRITEBACK_sum" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_WR_UNCACHED_32B_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_NORMAL_WRITEBACK_sum"; do
  rm -rf /tmp/sp_pmc
  timeout 120 rocprofv3 --pmc $set --output-format csv -d /tmp/sp_pmc -- $R/tools/store_pattern strips > /dev/null 2>&1
  f=$(find /tmp/sp_pmc -name "*counter_collection.csv" | head -1)
  python3 - $f <<'PY'
import csv, sys
from collections import defaultdict
rows = defaultdict(dict)
for r in csv.DictReader(open(sys.argv[1])):
    if "k_strips" in r["Kernel_Name"]:
        rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(rows)
names = ["raster17", "grid5", "grid4", "raster16"]
for m in range(4):
    sel = ids[m * 6 + 1:(m + 1) * 6]       # skip the warm-up dispatch of each mode
    if not sel: continue
    keys = sorted(rows[sel[0]])
    print(names[m], {k: round(sum(rows[i][k] for i in sel) / len(sel)) for k in keys})
PY
done 2>&1 | tee $O/strips_pmc.log
