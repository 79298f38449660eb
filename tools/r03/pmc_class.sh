#!/bin/bash
# Utilisation counters of the per-class kernels of one workload (BEVW_PLAN_ONELAUNCH=0): gpurun -- 'bash tools/r03/pmc_class.sh [workload] [tag] [onelaunch]'
W=${1:-direct_stitch_b256}
TAG=${2:-pmc}
ONE=${3:-0}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${TAG}_$W
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
while read -r set; do
  [ -z "$set" ] && continue
  i=$((i+1))
  rm -rf /tmp/pmc_$i
  BEVW_PLAN_ONELAUNCH=$ONE timeout 90 rocprofv3 --pmc $set --output-format csv -d /tmp/pmc_$i -- python $R/bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline > /tmp/pmc_$i.log 2>&1
  f=$(find /tmp/pmc_$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then cp $f $O/pass_$i.csv; else echo "pass $i FAILED: $set"; tail -2 /tmp/pmc_$i.log; fi
done <<'SETS'
GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES
SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS
SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum
TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
TCP_TCC_WRITE_REQ_sum TCP_TCR_TCP_STALL_CYCLES_sum
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
SETS
cd $R
python - "$O" <<'PY' | tee $O/summary.txt
import csv, glob, sys
from collections import defaultdict
tot = defaultdict(lambda: defaultdict(float)); n = defaultdict(lambda: defaultdict(int))
for f in sorted(glob.glob(sys.argv[1] + "/pass_*.csv")):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if "k_plan_" in k and "build" not in k and "touch" not in k:
            k = k.split("(")[0].replace("void bevw::", "")
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
names = sorted({c for k in tot for c in tot[k]})
print("%-42s" % "counter" + "".join("%18s" % k[-16:] for k in tot))
for c in names:
    print("%-42s" % c + "".join("%18.0f" % (tot[k][c] / max(1, n[k][c])) for k in tot))
PY
