#!/bin/bash
# Utilisation counters of the per-frame stitch kernels (k_plan_block, k_plan_all, k_plan_lean) of one workload: gpurun -- 'bash tools/collect_pmc.sh [workload]'
# One rocprofv3 --pmc pass per line (small sets only: larger ones exceed the counter hardware and hang), each under a timeout.
W=${1:-direct_stitch_b256}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc_$W
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
while read -r set; do
  [ -z "$set" ] && continue
  i=$((i+1))
  rm -rf /tmp/pmc_$i
  timeout 90 rocprofv3 --pmc $set --output-format csv -d /tmp/pmc_$i -- python $R/bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline > /tmp/pmc_$i.log 2>&1
  f=$(find /tmp/pmc_$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then cp $f $O/pass_$i.csv; echo "pass $i ok: $set"; else echo "pass $i FAILED: $set"; tail -2 /tmp/pmc_$i.log; fi
done <<'SETS'
GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES
SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS
SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM
SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT
TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum
TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum
SETS
cd $R
python - "$O" <<'PY'
import csv, glob, sys
from collections import defaultdict
tot = defaultdict(float); n = defaultdict(int)
for f in sorted(glob.glob(sys.argv[1] + "/pass_*.csv")):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if "k_plan_all" in k or "k_plan_block" in k or "k_plan_lean" in k:
            k = k.split("(")[0].replace("void bevw::", "")
            tot[(k, r["Counter_Name"])] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
print("kernel, counter, per-launch value (the per-frame stitch kernels of one step)")
for k in sorted(tot):
    print("%-36s %-36s %16.0f   (%d dispatches)" % (k[0], k[1], tot[k] / max(1, n[k]), n[k]))
PY
