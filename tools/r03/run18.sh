#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_run18
mkdir -p $O
cd $R
timeout 1500 python tools/ab_bench.py --workload direct_stitch_b256 --reps 3 --steps 20 \
  base: spatial:BEVW_PLAN_SPATIAL=1 skew64:BEVW_UNIT_SKEW=64 skew32:BEVW_UNIT_SKEW=32 sc3:BEVW_UNIT_SECTOR_COST=3 \
  spatial_sc3:BEVW_PLAN_SPATIAL=1,BEVW_UNIT_SECTOR_COST=3 spatial_skew64:BEVW_PLAN_SPATIAL=1,BEVW_UNIT_SKEW=64 > $O/ab_direct.log 2>&1; cat $O/ab_direct.log
timeout 300 env BEVW_PLAN_SPATIAL=1 BEVW_UNIT_SKEW=64 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
for v in base spatial skew64; do
  case $v in base) E="";; spatial) E="BEVW_PLAN_SPATIAL=1";; skew64) E="BEVW_UNIT_SKEW=64";; esac
  rm -rf /tmp/pw_$v
  env $E timeout 120 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum --output-format csv -d /tmp/pw_$v -- python $R/bench.py --workload direct_stitch_b256 --steps 3 --warmup 1 --placements 1 --no-cpu-baseline > /tmp/pw_$v.log 2>&1
  f=$(find /tmp/pw_$v -name "*counter_collection.csv" | head -1)
  python - $f $v <<'PY'
import csv,sys
from collections import defaultdict
t=defaultdict(float);n=defaultdict(int)
for r in csv.DictReader(open(sys.argv[1])):
    if "k_plan_all" in r["Kernel_Name"]:
        t[r["Counter_Name"]]+=float(r["Counter_Value"]);n[r["Counter_Name"]]+=1
print(sys.argv[2], {k:round(t[k]/n[k]) for k in t})
PY
done 2>&1 | tee $O/write_counters.log
