#!/bin/bash
# conflict-free gain LUT: GPU suite, k_gain_lut time old / new build (interleaved kernel traces), config 4 A/B
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02_run29
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; grep -n "passed\|failed\|rror" $O/pytest.log | tail -5
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
for v in base new; do
  rm -rf /tmp/kt
  if [ $v = base ]; then export BEVW_LIB_PATH=$R/build_var/libbevwarp_r02base.so; else unset BEVW_LIB_PATH; fi
  timeout 180 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $R/bench.py --workload blend_balance_b256 --steps 10 --warmup 2 --no-cpu-baseline > /tmp/kt.log 2>&1
  python - $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $v <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if any(k in r["Name"] for k in ("k_gain", "k_vsum", "k_lum_groups", "k_plan_all", "k_plan_block")):
        print("   %s %-50s avg %8.1f us" % (sys.argv[2], r["Name"].split("(")[0][-50:], float(r["AverageNs"]) / 1e3))
PY
done
done
unset BEVW_LIB_PATH
cd $R
python tools/ab_bench.py --workload blend_balance_b256 --reps 5 base:BEVW_LIB_PATH=build_var/libbevwarp_r02base.so new:X=1 2>&1 | tee $O/ab.log
