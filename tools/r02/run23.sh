#!/bin/bash
# block-staged classes on a second stream (fork / join): A/B
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02_run23
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; grep -n "passed\|failed\|rror" $O/pytest.log | tail -5
B="python bench.py --steps 40 --warmup 5 --no-cpu-baseline --workload"
res() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('RESULT $1 ms %.4f median %.4f frac %.4f' % (d['roofline']['kernel_ms'], d['roofline'].get('kernel_ms_median', 0), d['roofline']['frac']))"; }
for rep in 1 2; do
for w in direct_stitch_b256 blend_b256 undistort_b64 blend_4k blend_balance_b256; do
  BEVW_PLAN_BLOCK=0 timeout 300 $B $w 2>&1 | tail -1 | res "$w block0" | tee -a $O/ab.log
  BEVW_PLAN_TWOSTREAMS=0 timeout 300 $B $w 2>&1 | tail -1 | res "$w block1 one stream" | tee -a $O/ab.log
  BEVW_PLAN_TWOSTREAMS=1 timeout 300 $B $w 2>&1 | tail -1 | res "$w block1 two streams" | tee -a $O/ab.log
done
done
