#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_run23
mkdir -p $O
cd $R
for i in 1 2; do
  timeout 600 python bench.py --workload direct_stitch_b256 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_$i.json
  python - $O/bench_$i.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
o=d.get("other_output_layout")
print(d["config"]["workload"], "ms/step %.4f frac %.3f placements %s | other(%s) %s" % (d["ms_per_step"], d["roofline"]["frac"], d["placements"]["ms_per_step"], o and o["output_layout"], o and ("%.4f frac %.3f %s" % (o["ms_per_step"], o["frac"], o["placements_ms_per_step"]))))
PY
done
