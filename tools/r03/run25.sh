#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_run25
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_analytic.py -m gpu -x -q 2>&1 | tail -3
for w in direct_stitch_analytic_f32_b64 direct_stitch_analytic_f64_b64; do
  timeout 600 python bench.py --workload $w --no-cpu-baseline --placements 3 2>/dev/null | tail -1 > $O/bench_$w.json
  python -c "import json;d=json.load(open('$O/bench_$w.json'));print('$w ms/step %.4f frac %.3f' % (d['ms_per_step'], d['roofline']['frac']))"
done
