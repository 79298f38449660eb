#!/bin/bash
# round 4: uniform per-wave walks in the tail of the synchronisation: JPEG tests, decode lines (synthetic / the reference's files), kernel stats
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_run11
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_jpeg_gpu.py tests/test_tools.py tests/test_jpeg_goldens.py -m gpu -x -q > $O/pytest_jpeg.log 2>&1; grep -n "passed\|failed" $O/pytest_jpeg.log | tail -1
for src in synthetic repo; do
  timeout 300 python bench.py --workload jpeg_decode_b64 --jpeg-source $src --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_jpeg_decode_b64_$src.json
  python -c "import json;d=json.load(open('$O/bench_jpeg_decode_b64_$src.json'));c=d['config'];print('$src decode',round(d['value']),'ms',round(d['ms_per_step'],3),'rounds',c.get('fixed_point_rounds_max'))"
done
timeout 300 python bench.py --workload jpeg_bev_jpeg_b64 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_jpeg_bev_jpeg_b64.json
python -c "import json;d=json.load(open('$O/bench_jpeg_bev_jpeg_b64.json'));c=d['config'];print('pipeline',round(d['value']),'ms',round(d['ms_per_step'],3),'host_api',c.get('host_api_frames_per_s'),c.get('host_api_unpipelined_frames_per_s'))"
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
    for r in list(csv.DictReader(open('gpurun_out/r04_run11/rocprofv3_kernel_stats_jpeg_decode_b64_%s.csv' % src)))[:7]:
        print('%-36s calls %4s avg_us %8.1f %6s%%' % (r['Name'].split('(')[0].replace('void ','').replace('bevw::jpg::','')[:36], r['Calls'], float(r['AverageNs'])/1e3, r['Percentage']))
P
BEVW_SOAK_SECONDS=60 timeout 120 python tools/soak_jpeg.py --cases 20000 --seed 13 2>&1 | grep -v Suspension | tail -1
