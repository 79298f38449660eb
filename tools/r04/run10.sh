#!/bin/bash
# round 4: extended soaks on the final build (time-boxed; every script reports how many cases it ran), the default bench line once more
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_run10
mkdir -p $O
cd $R
( time timeout 900 python bench.py ) > $O/bench_default.json 2> $O/bench_default.time; tail -3 $O/bench_default.time | head -1
python -c "
import json
d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1])
print('default', round(d['value']), 'ms', round(d['ms_per_step'],4), 'frac', round(d['roofline']['frac'],3), 'other', round(d['other_output_layout']['ms_per_step'],4), {k:(round(v['value']) if isinstance(v,dict) and 'value' in v else None) for k,v in d['f4'].items()})"
BEVW_SOAK_SECONDS=330 timeout 420 python tools/soak_stitch.py 4000 20000 > $O/soak_stitch_4000.log 2>&1; tail -2 $O/soak_stitch_4000.log
BEVW_SOAK_SECONDS=120 timeout 200 python tools/soak_analytic.py 1500 9000 > $O/soak_analytic_1500.log 2>&1; tail -2 $O/soak_analytic_1500.log
BEVW_SOAK_SECONDS=150 timeout 240 python tools/soak_jpeg.py --cases 20000 --seed 11 2>&1 | grep -v Suspension > $O/soak_jpeg_seed11.log; tail -2 $O/soak_jpeg_seed11.log
timeout 150 python tools/soak_warps.py 600 > $O/soak_warps.log 2>&1; tail -2 $O/soak_warps.log
