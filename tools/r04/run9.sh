#!/bin/bash
# round 4: after the lazy second engine stream + f4 legs in their own processes: GPU suite, the default line, the JPEG lines with the VALU roofline, hardware-queue count A/B
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_run9
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; grep -n "passed\|failed" $O/pytest_gpu.log | tail -1
( time timeout 900 python bench.py ) > $O/bench_default.json 2> $O/bench_default.time; tail -3 $O/bench_default.time | head -1
python - <<'P'
import json
d=json.loads(open('gpurun_out/r04_run9/bench_default.json').read().strip().splitlines()[-1])
print('default', round(d['value']), 'ms', round(d['ms_per_step'],4), 'frac', round(d['roofline']['frac'],3), 'other', d['other_output_layout'] and round(d['other_output_layout']['ms_per_step'],4))
for k,v in (d.get('f4') or {}).items():
    if isinstance(v, dict) and 'value' in v: print('  f4', k, round(v['value']), v['unit'], 'ms', round(v['ms_per_step'],3), v['roofline']['bound'], round(v['roofline']['frac'],3), 'host_api', v.get('host_api_frames_per_s'), 'cpu', v.get('cpu_baseline') and round(v['cpu_baseline']['value']))
    else: print('  f4', k, str(v)[:300])
P
for w in jpeg_decode_b64 jpeg_encode_b64 jpeg_bev_jpeg_b64; do
  timeout 600 python bench.py --workload $w 2>/dev/null | tail -1 > $O/bench_$w.json
  python -c "import json;d=json.load(open('$O/bench_$w.json'));c=d['config'];print('$w',round(d['value']),d['unit'],'ms',round(d['ms_per_step'],3),'frac',round(d['roofline']['frac'],3),d['roofline']['bound'],'host_api',c.get('host_api_frames_per_s'),c.get('host_api_unpipelined_frames_per_s'),'| cpu',d['cpu_baseline'] and round(d['cpu_baseline']['value'],1))"
done
timeout 600 python bench.py --workload jpeg_decode_b64 --jpeg-source repo 2>/dev/null | tail -1 > $O/bench_jpeg_decode_b64_repo_files.json
python -c "import json;d=json.load(open('$O/bench_jpeg_decode_b64_repo_files.json'));print('decode repo',round(d['value']),'ms',round(d['ms_per_step'],3),'frac',round(d['roofline']['frac'],3),d['roofline']['bound'])"
for q in 2 4 8 16; do
  GPU_MAX_HW_QUEUES=$q timeout 300 python bench.py --workload jpeg_bev_jpeg_b64 --no-cpu-baseline 2>/dev/null | tail -1 > $O/q$q.json
  python -c "import json;d=json.load(open('$O/q$q.json'));c=d['config'];print('GPU_MAX_HW_QUEUES=$q pipeline',round(d['value']),'ms',round(d['ms_per_step'],3),'host_api',c.get('host_api_frames_per_s'),c.get('host_api_unpipelined_frames_per_s'))"
done
bash tools/r04/ab.sh hwq "blend_balance_b256" 2 12 "--placements 1 --single-layout" q4: q8:GPU_MAX_HW_QUEUES=8 q2:GPU_MAX_HW_QUEUES=2
