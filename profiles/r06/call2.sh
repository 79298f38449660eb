#!/bin/bash
# GPU call 2 of round 6: RCCL stand-in parity (world 2 / 4 on one GPU), ring-buffer parity, copy yardstick, write-request counters of the
# two store formats, config-4 slice sweep (with and without ring buffers), cut-cost sweep of the unit compiler
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_call2
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_camera_shard.py -m gpu -q -x -k "rccl" -s > $O/pytest_rccl.log 2>&1; tail -5 $O/pytest_rccl.log
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "balance_slices or copy_yardstick or blend" > $O/pytest_new.log 2>&1; tail -3 $O/pytest_new.log
python - <<'PY' 2>&1 | tee $O/copy_rate.log
import ctypes as C
from cameracalibration_amd import _ffi
L = _ffi.lib()
for streaming in (0, 1):
    for nbytes in (1 << 30, 2 << 30):
        v = C.c_double()
        _ffi.check(L.bevw_device_copy_rate(0, nbytes, 10, streaming, C.byref(v)))
        print("copy yardstick: %d MB streaming=%d  %.1f GB/s moved" % (nbytes >> 20, streaming, v.value))
PY
V=build_var
AB="python tools/ab_bench.py --reps 3 --steps 20"
timeout 900 $AB --workload blend_balance_b256 --bench-args "--placements 2 --single-layout" p2: p4:BEVW_BAL_PARTS=4 p8:BEVW_BAL_PARTS=8 p16:BEVW_BAL_PARTS=16 p32:BEVW_BAL_PARTS=32 \
   p4r:BEVW_BAL_PARTS=4,BEVW_BAL_RING=1 p8r:BEVW_BAL_PARTS=8,BEVW_BAL_RING=1 p16r:BEVW_BAL_PARTS=16,BEVW_BAL_RING=1 p32r:BEVW_BAL_PARTS=32,BEVW_BAL_RING=1 \
   p16r_nb8:BEVW_BAL_PARTS=16,BEVW_BAL_RING=1,BEVW_PLAN_NB=8 p32r_nb8:BEVW_BAL_PARTS=32,BEVW_BAL_RING=1,BEVW_PLAN_NB=8 2>&1 | tee -a $O/ab.log
timeout 900 $AB --workload direct_stitch_b256 --bench-args "--placements 2 --single-layout" base: l2s3:BEVW_UNIT_SECTOR_COST=3 l2s1:BEVW_UNIT_SECTOR_COST=1 l3s2:BEVW_UNIT_LINE_COST=3,BEVW_UNIT_SECTOR_COST=2 \
   l4s2:BEVW_UNIT_LINE_COST=4,BEVW_UNIT_SECTOR_COST=2 l1s8:BEVW_UNIT_LINE_COST=1,BEVW_UNIT_SECTOR_COST=8 l2s16:BEVW_UNIT_SECTOR_COST=16 \
   loads_only:BEVW_LIB_PATH=$V/libbevwarp_x2.so stores_only:BEVW_LIB_PATH=$V/libbevwarp_x3.so mem:BEVW_LIB_PATH=$V/libbevwarp_x1.so 2>&1 | tee -a $O/ab.log
# write / read request counters of the two store formats (one small --pmc set per pass, no tracing)
cd /tmp && export TMPDIR=/tmp
for v in s0 s1; do
  E=""; [ $v = s1 ] && E="BEVW_LIB_PATH=$R/$V/libbevwarp_s1.so"
  i=0
  for set in "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum" "SQ_INSTS_VMEM_WR SQ_INSTS_VALU TA_TA_BUSY_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_REQ_sum"; do
    i=$((i+1)); rm -rf /tmp/pm_${v}_$i
    env $E timeout 120 rocprofv3 --pmc $set --output-format csv -d /tmp/pm_${v}_$i -- python $R/bench.py --workload direct_stitch_b256 --steps 3 --warmup 1 --placements 1 --single-layout --no-cpu-baseline --no-f4 --no-live-traffic > /tmp/pm_${v}_$i.log 2>&1
    f=$(find /tmp/pm_${v}_$i -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f $O/pmc_${v}_pass$i.csv || { echo "pass $v $i failed"; tail -3 /tmp/pm_${v}_$i.log; }
  done
done
python - $O <<'PY' | tee $O/pmc_store_format.txt
import csv, glob, sys
from collections import defaultdict
for v in ("s0", "s1"):
    t = defaultdict(float); n = defaultdict(int)
    for f in sorted(glob.glob(sys.argv[1] + "/pmc_%s_pass*.csv" % v)):
        for r in csv.DictReader(open(f)):
            if "k_plan_units" in r["Kernel_Name"]:
                t[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    print(v, "(12-byte stores)" if v == "s0" else "(repacked 16-byte stores)")
    for c in sorted(t): print("   %-36s %16.0f per launch" % (c, t[c] / max(1, n[c])))
PY
