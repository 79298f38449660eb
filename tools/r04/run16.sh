#!/bin/bash
# after the store policy change: parity subset, the stitch lines (both layouts where they exist), the default line
R=$(pwd); O=$R/gpurun_out/r04_run16; mkdir -p $O
timeout 300 python -m pytest tests -m gpu -x -q -k "bench_configuration or repo_data or analytic or undistort or camera_methods or pitched" 2>&1 | tail -1 | tee $O/pytest_subset.log
for w in blend_b256 undistort_b64; do
  timeout 200 python bench.py --workload $w --no-f4 --no-cpu-baseline --placements 3 2>/dev/null | tail -1 > $O/bench_$w.json
  python -c "import json;d=json.load(open('$O/bench_$w.json'));o=d.get('other_output_layout');print('$w',round(d['value']),'ms',round(d['ms_per_step'],4),'frac',round(d['roofline']['frac'],3),'| other',o and round(o['ms_per_step'],4))"
done
( time timeout 600 python bench.py ) > $O/bench_default.json 2> $O/bench_default.time
python -c "
import json
d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1])
print('default',round(d['value']),'ms',round(d['ms_per_step'],4),'frac',round(d['roofline']['frac'],3),'dense',round(d['other_output_layout']['ms_per_step'],4),d['placements']['ms_per_step'])"
timeout 200 python bench.py --workload direct_stitch_b256 --no-f4 2>/dev/null | tail -1 > $O/bench_direct_stitch_b256.json
python -c "import json;d=json.load(open('$O/bench_direct_stitch_b256.json'));o=d.get('other_output_layout');print('direct_stitch_b256',round(d['value']),'ms',round(d['ms_per_step'],4),'frac',round(d['roofline']['frac'],3),d['placements']['ms_per_step'],'| other',o and round(o['ms_per_step'],4))"
cd /tmp; export TMPDIR=/tmp
for w in direct_stitch_b256; do
  rm -rf /tmp/kt_$w
  timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$w -- python $R/bench.py --workload $w --steps 10 --warmup 2 --placements 1 --single-layout --no-cpu-baseline --no-f4 > /tmp/kt_$w.log 2>&1
  cp $(find /tmp/kt_$w -name "*kernel_stats.csv" | head -1) $O/rocprofv3_kernel_stats_$w.csv
done
