#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02_run9
mkdir -p $O
cd $R
echo "== pytest (spatial)"; timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee $O/pytest.log
B="python bench.py --steps 40 --warmup 5 --no-cpu-baseline --workload"
res() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('RESULT $1 ms %.4f frac %.4f' % (d['roofline']['kernel_ms'], d['roofline']['frac']))"; }
for rep in 1 2; do
for ol in 1 2; do
  BEVW_PLAN_ONELAUNCH=$ol timeout 300 $B direct_stitch_b256 2>&1 | tail -1 | res "direct onelaunch$ol" | tee -a $O/ab.log
done
done
for nb in 4 16 32; do BEVW_PLAN_NB=$nb timeout 300 $B direct_stitch_b256 2>&1 | tail -1 | res "direct spatial nb$nb" | tee -a $O/ab.log; done
for w in blend_b256 undistort_b64 blend_4k blend_balance_b256; do
  for ol in 1 2; do BEVW_PLAN_ONELAUNCH=$ol timeout 300 $B $w 2>&1 | tail -1 | res "$w onelaunch$ol" | tee -a $O/ab.log; done
done
