#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02_run7
mkdir -p $O
cd $R
echo "== pytest (coop store)"; timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee $O/pytest_pair.log
B="python bench.py --steps 40 --warmup 5 --no-cpu-baseline --workload"
res() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('RESULT $1 ms %.4f frac %.4f' % (d['roofline']['kernel_ms'], d['roofline']['frac']))"; }
for rep in 1 2; do
for coop in 0 1; do
  BEVW_PAIR_COOP=$coop timeout 300 $B direct_stitch_b256 2>&1 | tail -1 | res "direct coop$coop" | tee -a $O/ab.log
done
done
BEVW_PAIR_COOP=1 BEVW_PLAN_COLMAJOR=0 timeout 300 $B direct_stitch_b256 2>&1 | tail -1 | res "direct coop1 colmajor0" | tee -a $O/ab.log
BEVW_PAIR_COOP=1 BEVW_PLAN_LX=16 timeout 300 $B direct_stitch_b256 2>&1 | tail -1 | res "direct coop1 lx16" | tee -a $O/ab.log
BEVW_PAIR_COOP=1 BEVW_PLAN_LX=4 timeout 300 $B direct_stitch_b256 2>&1 | tail -1 | res "direct coop1 lx4" | tee -a $O/ab.log
BEVW_PAIR_COOP=1 BEVW_PLAN_NB=16 timeout 300 $B direct_stitch_b256 2>&1 | tail -1 | res "direct coop1 nb16" | tee -a $O/ab.log
for w in blend_b256 undistort_b64 blend_4k blend_balance_b256; do
  for coop in 0 1; do BEVW_PAIR_COOP=$coop timeout 300 $B $w 2>&1 | tail -1 | res "$w coop$coop" | tee -a $O/ab.log; done
done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt1; BEVW_PLAN_ONELAUNCH=0 timeout 180 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt1 -- python $R/bench.py --workload direct_stitch_b256 --steps 10 --warmup 2 --no-cpu-baseline > /tmp/kt1.log 2>&1
cp $(find /tmp/kt1 -name "*kernel_stats.csv" | head -1) $O/kernel_stats_per_class.csv; head -9 $O/kernel_stats_per_class.csv | cut -c1-150
