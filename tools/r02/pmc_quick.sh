#!/bin/bash
# a few counter passes of the merged per-frame kernel for several build / env variants: label|env assignments|lib
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02_pmcq
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() {  # label, lib, extra env...
  label=$1; lib=$2; shift 2
  i=0
  while read -r set; do
    [ -z "$set" ] && continue
    i=$((i+1)); rm -rf /tmp/pq_$i
    env "$@" BEVW_LIB_PATH=$lib timeout 90 rocprofv3 --pmc $set --output-format csv -d /tmp/pq_$i -- python $R/bench.py --workload direct_stitch_b256 --steps 3 --warmup 1 --no-cpu-baseline > /tmp/pq_$i.log 2>&1
    f=$(find /tmp/pq_$i -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && cp $f $O/${label}_pass_$i.csv || { echo "pass $i FAILED ($label)"; tail -2 /tmp/pq_$i.log; }
  done <<'SETS'
GRBM_GUI_ACTIVE TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum
TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum
TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum
TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum
TCC_WRITEBACK_sum TCC_NORMAL_EVICT_sum TCC_TAG_STALL_sum
SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY
SETS
  python - "$O" "$label" <<'PY'
import csv, glob, sys
from collections import defaultdict
tot = defaultdict(float); n = defaultdict(int)
for f in sorted(glob.glob(sys.argv[1] + "/" + sys.argv[2] + "_pass_*.csv")):
    for r in csv.DictReader(open(f)):
        if "k_plan_all" in r.get("Kernel_Name", ""):
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
print("== " + sys.argv[2])
for c in tot: print("   %-36s %14.0f" % (c, tot[c] / max(1, n[c])))
PY
}
L=$R/cameracalibration_amd/libbevwarp.so
run coop0 $L BEVW_PAIR_COOP=0
run coop1 $L BEVW_PAIR_COOP=1
run abl7 $R/build_abl/libbevwarp_abl7.so BEVW_PAIR_COOP=0
