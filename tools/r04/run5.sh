#!/bin/bash
# round 4, fifth GPU call: whole GPU suite after the revert of the grouped columns + the per-tap kernel's register budget; JPEG slices A/B; stitch frames-per-block A/B
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_run5
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; grep -n "passed\|failed" $O/pytest_gpu.log | tail -2
for v in parts2: parts3:BEVW_JPEG_PARTS=3 parts4:BEVW_JPEG_PARTS=4 parts2b: parts1:BEVW_JPEG_PARTS=1; do
  n=${v%%:*}; e=${v#*:}
  env $e timeout 300 python bench.py --workload jpeg_decode_b64 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_jpeg_decode_b64_$n.json
  python -c "import json;d=json.load(open('$O/bench_jpeg_decode_b64_$n.json'));c=d['config'];print('$n decode',round(d['value']),'ms',round(d['ms_per_step'],3))"
  env $e timeout 300 python bench.py --workload jpeg_decode_b64 --jpeg-source repo --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_jpeg_decode_b64_repo_$n.json
  python -c "import json;d=json.load(open('$O/bench_jpeg_decode_b64_repo_$n.json'));c=d['config'];print('$n decode repo',round(d['value']),'ms',round(d['ms_per_step'],3))"
done
bash tools/r04/ab.sh stitch "direct_stitch_b256" 2 20 "--placements 2 --single-layout" base: nb32:BEVW_PLAN_NB=32 nb8:BEVW_PLAN_NB=8 base2:
bash tools/r04/ab.sh stitch_dense "direct_stitch_b256" 2 20 "--placements 2 --single-layout --output-pitch dense" base: nb32:BEVW_PLAN_NB=32 skew64:BEVW_UNIT_SKEW=64
