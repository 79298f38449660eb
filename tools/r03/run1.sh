#!/bin/bash
# round 3, first GPU run of the unit schedule: parity, A/B against round 2's schedule, per-class kernel times
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_run1
mkdir -p $O
cd $R
timeout 420 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/pytest_parity.log 2>&1; tail -3 $O/pytest_parity.log
timeout 400 python tools/ab_bench.py --workload direct_stitch_b256 --reps 3 --steps 30 \
  r02:BEVW_PLAN_UNITS=0 units_w3: units_w4:BEVW_LIB_PATH=$R/build_var/libbevwarp_w4.so > $O/ab_direct.log 2>&1; cat $O/ab_direct.log
timeout 200 python tools/ab_bench.py --workload undistort_b64 --reps 2 --steps 30 r02:BEVW_PLAN_UNITS=0 units_w3: > $O/ab_undistort.log 2>&1; cat $O/ab_undistort.log
timeout 200 python tools/ab_bench.py --workload blend_b256 --reps 2 --steps 30 r02:BEVW_PLAN_UNITS=0 units_w3: > $O/ab_blend.log 2>&1; cat $O/ab_blend.log
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt_classes
BEVW_PLAN_ONELAUNCH=0 timeout 180 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_classes -- python $R/bench.py --workload direct_stitch_b256 --steps 10 --warmup 2 --no-cpu-baseline > /tmp/kt_classes.log 2>&1
cp $(find /tmp/kt_classes -name "*kernel_stats.csv" | head -1) $O/kernel_stats_per_class.csv
rm -rf /tmp/kt_one
timeout 180 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_one -- python $R/bench.py --workload direct_stitch_b256 --steps 10 --warmup 2 --no-cpu-baseline > /tmp/kt_one.log 2>&1
cp $(find /tmp/kt_one -name "*kernel_stats.csv" | head -1) $O/kernel_stats_merged.csv
head -12 $O/kernel_stats_per_class.csv | cut -c1-160
head -4 $O/kernel_stats_merged.csv | cut -c1-160
