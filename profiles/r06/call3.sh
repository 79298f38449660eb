#!/bin/bash
# GPU call 3 of round 6: full GPU suite on the new default build, block timelines, the analytic lines, FETCH/WRITE of config 4 in 8 slices with / without ring buffers
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_call3
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
X4=build_var/libbevwarp_x4.so
for v in "" "--blend" "--env BEVW_PLAN_NB=8" "--env BEVW_PLAN_NB=32" "--env BEVW_PLAN_XCDMAP=2" "--dense"; do
  echo "=== block_timeline $v" | tee -a $O/timeline.log
  BEVW_LIB_PATH=$X4 timeout 300 python tools/block_timeline.py $v 2>&1 | tee -a $O/timeline.log
done
for w in direct_stitch_analytic_perpixel_b64 direct_stitch_analytic_f32_b64; do
  timeout 300 python bench.py --workload $w --steps 10 --warmup 2 --placements 3 --no-cpu-baseline --no-f4 > $O/bench_$w.json 2> $O/bench_$w.err; python - $O/bench_$w.json <<'PY'
import json, sys
o = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(o["config"]["workload"], o["value"], o["unit"], "ms", o["ms_per_step"], "frac", o["roofline"]["frac"], "traffic x", o["roofline"]["traffic_over_algorithmic"], o["roofline"]["traffic_source"])
PY
done
cd /tmp && export TMPDIR=/tmp
for v in p8 p8r; do
  E="BEVW_BAL_PARTS=8"; [ $v = p8r ] && E="BEVW_BAL_PARTS=8 BEVW_BAL_RING=1"
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pm_${v}_$c
    env $E timeout 200 rocprofv3 --pmc $c --output-format csv -d /tmp/pm_${v}_$c -- python $R/bench.py --workload blend_balance_b256 --steps 3 --warmup 1 --placements 1 --single-layout --no-cpu-baseline --no-f4 --no-live-traffic > /tmp/pm_${v}_$c.log 2>&1
    f=$(find /tmp/pm_${v}_$c -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f $O/pmc_c4_${v}_$c.csv || { echo "pass $v $c failed"; tail -3 /tmp/pm_${v}_$c.log; }
  done
done
python - $O <<'PY' | tee $O/pmc_c4_slices.txt
import csv, glob, sys
from collections import defaultdict
for v in ("p8", "p8r"):
    print(v, "(8 slices)" if v == "p8" else "(8 slices, ring buffers)")
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        t = defaultdict(float)
        try:
            for r in csv.DictReader(open("%s/pmc_c4_%s_%s.csv" % (sys.argv[1], v, c))):
                k = r["Kernel_Name"].split("(")[0].replace("void bevw::", "")
                t[k] += float(r["Counter_Value"])
        except OSError:
            continue
        for k in sorted(t):
            if any(s in k for s in ("k_vsum", "k_lum", "k_plan_units", "k_gain", "k_stitch_plan")):
                print("   %-12s %-40s %10.1f MB per step (raw counter KB x 1024 / 4 steps, uncorrected)" % (c, k[-40:], t[k] * 1024 / 4 / 1e6))
PY
