#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_run22
mkdir -p $O
cd $R
for w in direct_stitch_b256 blend_b256 blend_balance_b256 blend_4k undistort_b64; do
  timeout 600 python bench.py --workload $w --no-cpu-baseline 2>$O/err_$w.log | tail -1 > $O/bench_$w.json
  python - $O/bench_$w.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
o=d.get("other_output_layout")
print(d["config"]["workload"], "ms/step %.4f frac %.3f placements %s | other(%s) %s" % (d["ms_per_step"], d["roofline"]["frac"], d["placements"]["ms_per_step"], o and o["output_layout"], o and ("%.4f frac %.3f %s" % (o["ms_per_step"], o["frac"], o["placements_ms_per_step"]))))
PY
done
