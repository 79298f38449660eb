#!/bin/bash
# round 4, fourth GPU call: whole GPU suite (grouped columns in the Huffman walker, the RCCL world-n test at world 1), JPEG bench lines + kernel stats
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_run4
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; grep -n "passed\|failed" $O/pytest_gpu.log | tail -2
for w in jpeg_decode_b64 jpeg_bev_jpeg_b64 jpeg_encode_b64; do
  timeout 300 python bench.py --workload $w --no-cpu-baseline 2>$O/bench_$w.err | tail -1 > $O/bench_$w.json
  python -c "import json;d=json.load(open('$O/bench_$w.json'));c=d['config'];print('$w',round(d['value']),d['unit'],'ms',round(d['ms_per_step'],3),'rounds',c.get('fixed_point_rounds_max'),'host_api',c.get('host_api_frames_per_s'),c.get('host_api_unpipelined_frames_per_s'))"
done
timeout 300 python bench.py --workload jpeg_decode_b64 --jpeg-source repo --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_jpeg_decode_b64_repo.json
python -c "import json;d=json.load(open('$O/bench_jpeg_decode_b64_repo.json'));c=d['config'];print('decode repo',round(d['value']),'ms',round(d['ms_per_step'],3),'rounds',c.get('fixed_point_rounds_max'))"
cd /tmp && export TMPDIR=/tmp
for src in synthetic repo; do
  rm -rf /tmp/kt_$src
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$src -- python $R/bench.py --workload jpeg_decode_b64 --jpeg-source $src --steps 6 --warmup 2 --no-cpu-baseline > /tmp/kt_$src.log 2>&1
  cp $(find /tmp/kt_$src -name "*kernel_stats.csv" | head -1) $O/rocprofv3_kernel_stats_jpeg_decode_b64_$src.csv
done
cd $R
python - <<'P'
import csv
for src in ('synthetic','repo'):
    print('==', src)
    for r in list(csv.DictReader(open('gpurun_out/r04_run4/rocprofv3_kernel_stats_jpeg_decode_b64_%s.csv' % src)))[:12]:
        print('%-36s calls %4s avg_us %8.1f %6s%%' % (r['Name'].split('(')[0].replace('void ','').replace('bevw::jpg::','')[:36], r['Calls'], float(r['AverageNs'])/1e3, r['Percentage']))
P
