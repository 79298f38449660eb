#!/bin/bash
# round 2, GPU call 3: pair schedule v2 (sliced sparse tiles, 2-deep prefetch): parity, one launch vs per class, ablations
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02_run3
mkdir -p $O
cd $R
echo "== pytest (pair v2)"; timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee $O/pytest_pair.log
echo "== pytest per-class launches"; BEVW_PLAN_ONELAUNCH=0 timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee $O/pytest_pair_perclass.log
B="python bench.py --steps 40 --warmup 5 --no-cpu-baseline --workload"
res() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('RESULT $1 ms %.4f frac %.4f tiles %s' % (d['roofline']['kernel_ms'], d['roofline']['frac'], d['config'].get('tiles')))"; }
for w in direct_stitch_b256 blend_b256 undistort_b64 blend_4k blend_balance_b256; do
  for rep in 1 2; do
    timeout 300 $B $w 2>&1 | tail -1 | res "$w one_launch" | tee -a $O/ab.log
    BEVW_PLAN_ONELAUNCH=0 timeout 300 $B $w 2>&1 | tail -1 | res "$w per_class" | tee -a $O/ab.log
  done
done
for n in 1 2 3 4 5; do
  BEVW_LIB_PATH=$R/build_abl/libbevwarp_abl$n.so timeout 300 $B direct_stitch_b256 2>&1 | tail -1 | res abl$n | tee -a $O/abl.log
done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt1; BEVW_PLAN_ONELAUNCH=0 timeout 180 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt1 -- python $R/bench.py --workload direct_stitch_b256 --steps 10 --warmup 2 --no-cpu-baseline > /tmp/kt1.log 2>&1
cp $(find /tmp/kt1 -name "*kernel_stats.csv" | head -1) $O/kernel_stats_per_class.csv; head -9 $O/kernel_stats_per_class.csv | cut -c1-150
rm -rf /tmp/kt2; BEVW_PLAN_ONELAUNCH=0 timeout 180 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt2 -- python $R/bench.py --workload blend_4k --steps 10 --warmup 2 --no-cpu-baseline > /tmp/kt2.log 2>&1
cp $(find /tmp/kt2 -name "*kernel_stats.csv" | head -1) $O/kernel_stats_per_class_4k.csv; head -9 $O/kernel_stats_per_class_4k.csv | cut -c1-150
