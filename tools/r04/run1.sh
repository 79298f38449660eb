#!/bin/bash
# round 4, first GPU call: the whole GPU suite on the retired-schedules build, then config 3 / 4 / 2 / blend / 4K bench lines and the config-4 slices A/B
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_run1
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
for w in direct_stitch_b256 blend_b256 blend_balance_b256 undistort_b64 blend_4k; do
  timeout 300 python bench.py --workload $w --placements 3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_$w.json
  python -c "import json;d=json.load(open('$O/bench_$w.json'));o=d.get('other_output_layout');print('$w',round(d['value']),'ms',round(d['ms_per_step'],4),'frac',round(d['roofline']['frac'],3),d['placements']['ms_per_step'],'| other',o and (o['output_layout'],round(o['ms_per_step'],4)))"
done
bash tools/r04/ab.sh config4 blend_balance_b256 2 12 "--placements 1 --single-layout" parts1:BEVW_BAL_PARTS=1 parts2:BEVW_BAL_PARTS=2 parts4:BEVW_BAL_PARTS=4 parts8:BEVW_BAL_PARTS=8 parts4_noskew:BEVW_BAL_PARTS=4,BEVW_BAL_SKEW=0 parts2_noskew:BEVW_BAL_PARTS=2,BEVW_BAL_SKEW=0
