#!/bin/bash
# Counters of the JPEG kernels (row f4): gpurun -- 'bash tools/r03/pmc_jpeg.sh'  ->  gpurun_out/pmc_jpeg/{decode,encode}.txt
# Separate --pmc passes (instruction mix / activity, cache requests, HBM bytes); averages per launch and kernel.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc_jpeg
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for W in jpeg_decode_b64 jpeg_encode_b64; do
  i=0; rm -rf $O/$W; mkdir -p $O/$W
  while read -r set; do
    [ -z "$set" ] && continue
    i=$((i+1)); rm -rf /tmp/pj_$i
    BEVW_JPEG_PARTS=1 timeout 120 rocprofv3 --pmc $set --output-format csv -d /tmp/pj_$i -- python $R/bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline > /tmp/pj_$i.log 2>&1
    f=$(find /tmp/pj_$i -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f $O/$W/pass_$i.csv || { echo "pass $i failed"; tail -3 /tmp/pj_$i.log; }
  done <<'SETS'
GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS
SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES
TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_HIT_sum TCC_MISS_sum
FETCH_SIZE
WRITE_SIZE
SETS
  python - $O/$W <<'PY' | tee $O/${W#jpeg_}.txt
import csv,glob,sys
from collections import defaultdict
t=defaultdict(lambda: defaultdict(float)); n=defaultdict(lambda: defaultdict(int))
for f in sorted(glob.glob(sys.argv[1]+"/pass_*.csv")):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "k_jpeg" in k or "k_jenc" in k:
            k=k.split("(")[0].replace("void ","").replace("bevw::jpg::","")
            t[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[k][r["Counter_Name"]]+=1
names=sorted({c for k in t for c in t[k]})
print("%-28s"%"counter (average per launch)"+"".join("%20s"%k[-18:] for k in t))
for c in names: print("%-28s"%c+"".join("%20.0f"%(t[k][c]/max(1,n[k][c])) for k in t))
PY
done
