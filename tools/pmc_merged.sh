#!/bin/bash
# counters of the merged stitch kernel for one workload / env: gpurun -- 'bash tools/pmc_merged.sh tag workload "ENV=.. ENV=.." [bench args]'
TAG=${1:-pmc}; W=${2:-direct_stitch_b256}; E=${3:-}; shift 3
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
while read -r set; do
  [ -z "$set" ] && continue
  i=$((i+1)); rm -rf /tmp/pm_$i
  env $E timeout 120 rocprofv3 --pmc $set --output-format csv -d /tmp/pm_$i -- python $R/bench.py --workload $W --steps 3 --warmup 1 --placements 1 --single-layout --no-cpu-baseline --no-f4 --no-live-traffic "$@" > /tmp/pm_$i.log 2>&1
  f=$(find /tmp/pm_$i -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f $O/pass_$i.csv || { echo "pass $i failed"; tail -3 /tmp/pm_$i.log; }
done <<'SETS'
GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS
SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum
TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_LATENCY_sum
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_32B_sum
SETS
python - $O <<'PY' | tee $O/summary.txt
import csv,glob,sys
from collections import defaultdict
t=defaultdict(lambda: defaultdict(float)); n=defaultdict(lambda: defaultdict(int))
for f in sorted(glob.glob(sys.argv[1]+"/pass_*.csv")):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "k_plan_" in k and "build" not in k and "touch" not in k or "k_gain" in k or "k_vsum" in k or "k_lum" in k or "k_remap" in k:
            k=k.split("(")[0].replace("void bevw::","")
            t[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[k][r["Counter_Name"]]+=1
names=sorted({c for k in t for c in t[k]})
print("%-36s"%"counter"+"".join("%22s"%k[-20:] for k in t))
for c in names: print("%-36s"%c+"".join("%22.0f"%(t[k][c]/max(1,n[k][c])) for k in t))
PY
