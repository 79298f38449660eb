#!/bin/bash
# analytic projection mode: GPU tests + bench lines (+ the per-pixel kernel alone for comparison: BEVW_ANALYTIC_UNITS=0)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_analytic
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_analytic.py -m gpu -x -q -s 2>&1 | tail -12 | tee $O/pytest.log
for w in direct_stitch_analytic_f32_b64 direct_stitch_analytic_f64_b64; do
  timeout 300 python bench.py --workload $w --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_$w.json
  python -c "import json;d=json.load(open('$O/bench_$w.json'));print('$w',round(d['value']),'ms',round(d['ms_per_step'],4),'frac',round(d['roofline']['frac'],3),d['placements']['ms_per_step'])"
done
BEVW_ANALYTIC_UNITS=0 timeout 300 python bench.py --workload direct_stitch_analytic_f32_b64 --no-cpu-baseline --placements 1 2>/dev/null | tail -1 | python -c "import json,sys;d=json.loads(sys.stdin.read());print('per-pixel kernel only: ms',round(d['ms_per_step'],4))"
