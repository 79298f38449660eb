#!/bin/bash
# Row f4 evidence, run on the GPU box from the repo root:  gpurun -- 'bash tools/r03/jpeg_profiles.sh'
# bench lines of the three JPEG workloads + rocprofv3 kernel stats of decode and encode; lands under gpurun_out/jpeg/.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/jpeg
mkdir -p $O
cd $R
for w in jpeg_decode_b64 jpeg_encode_b64 jpeg_bev_jpeg_b64; do
  timeout 300 python bench.py --workload $w ${BENCH_ARGS:-} 2>$O/bench_$w.err | tail -1 > $O/bench_$w.json
  python -c "import json;d=json.load(open('$O/bench_$w.json'));print('$w',round(d['value']),d['unit'],'ms',round(d['ms_per_step'],4),'frac',round(d['roofline']['frac'],4),'| cpu',d['cpu_baseline'] and (round(d['cpu_baseline']['value'],1),d['cpu_baseline']['cores']),d['config'].get('fixed_point_rounds_max'))" || tail -5 $O/bench_$w.err
done
cd /tmp && export TMPDIR=/tmp
for w in jpeg_decode_b64 jpeg_encode_b64; do
  rm -rf /tmp/kt_$w
  timeout 180 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$w -- python $R/bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline > /tmp/kt_$w.log 2>&1
  cp $(find /tmp/kt_$w -name "*kernel_stats.csv" | head -1) $O/rocprofv3_kernel_stats_$w.csv
  head -12 $O/rocprofv3_kernel_stats_$w.csv | cut -c1-160
done
