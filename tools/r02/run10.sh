#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02_run10
mkdir -p $O
cd $R
B="python bench.py --steps 40 --warmup 5 --no-cpu-baseline --workload"
res() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('RESULT $1 ms %.4f frac %.4f' % (d['roofline']['kernel_ms'], d['roofline']['frac']))"; }
for rep in 1 2; do
  timeout 300 $B direct_stitch_b256 2>&1 | tail -1 | res "direct d2 (product)" | tee -a $O/ab.log
  for v in d4 d8 d44 d84; do
    BEVW_LIB_PATH=$R/build_var/libbevwarp_$v.so timeout 300 $B direct_stitch_b256 2>&1 | tail -1 | res "direct $v" | tee -a $O/ab.log
  done
done
for v in d8 d84; do
  BEVW_LIB_PATH=$R/build_var/libbevwarp_$v.so BEVW_PLAN_NB=16 timeout 300 $B direct_stitch_b256 2>&1 | tail -1 | res "direct $v nb16" | tee -a $O/ab.log
  BEVW_LIB_PATH=$R/build_var/libbevwarp_$v.so BEVW_PLAN_ONELAUNCH=1 timeout 300 $B direct_stitch_b256 2>&1 | tail -1 | res "direct $v onelaunch1" | tee -a $O/ab.log
done
BEVW_LIB_PATH=$R/build_var/libbevwarp_d84.so timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
for v in d8; do
rm -rf /tmp/kt1; BEVW_LIB_PATH=$R/build_var/libbevwarp_$v.so BEVW_PLAN_ONELAUNCH=0 timeout 180 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt1 -- python $R/bench.py --workload direct_stitch_b256 --steps 10 --warmup 2 --no-cpu-baseline > /tmp/kt1.log 2>&1
cp $(find /tmp/kt1 -name "*kernel_stats.csv" | head -1) $O/kernel_stats_per_class_$v.csv; head -6 $O/kernel_stats_per_class_$v.csv | cut -c1-150
done
