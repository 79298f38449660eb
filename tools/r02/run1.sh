#!/bin/bash
# round 2, GPU call 1: parity of the pair-staged schedule, A/B against the sector-staged one, microbenchmarks
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02_run1
mkdir -p $O
cd $R
echo "== pytest (pair-staged default)"; timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee $O/pytest_pair.log
echo "== pytest (sector-staged, BEVW_PLAN_STAGED=1)"; BEVW_PLAN_STAGED=1 timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee $O/pytest_sector.log
for w in direct_stitch_b256 blend_b256 undistort_b64 blend_4k; do
  for st in 1 2 1 2; do
    echo "== $w STAGED=$st"
    BEVW_PLAN_STAGED=$st timeout 300 python bench.py --workload $w --steps 40 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('RESULT $w staged=$st ms %.4f value %.0f frac %.4f tiles %s' % (d['roofline']['kernel_ms'], d['value'], d['roofline']['frac'], d['config'].get('tiles')))" | tee -a $O/ab.log
  done
done
echo "== hbm_stream"; timeout 120 tools/hbm_stream 2>&1 | tee $O/hbm_stream.log
echo "== valu_rates"; timeout 120 tools/valu_rates 2>&1 | tee $O/valu_rates.log
