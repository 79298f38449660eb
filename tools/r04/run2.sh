#!/bin/bash
# round 4, second GPU call: the GPU suite (new JPEG tests, pitched default layout, DPP channel sums), the default bench line with its f4 summary,
# the JPEG workloads + their VALU instruction counts, config 3 / 4 quick lines
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_run2
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log | head -2
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
( time timeout 600 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err; tail -4 $O/bench_default.err
python - <<'P'
import json
d=json.loads(open('gpurun_out/r04_run2/bench_default.json').read().strip().splitlines()[-1])
print('default', round(d['value']), 'ms', round(d['ms_per_step'],4), 'frac', round(d['roofline']['frac'],3), 'other', d['other_output_layout'] and round(d['other_output_layout']['ms_per_step'],4))
f=d.get('f4') or {}
for k,v in f.items():
    if isinstance(v, dict): print('  f4', k, round(v.get('value',0)), v.get('unit'), 'ms', round(v.get('ms_per_step',0),3), 'rounds', v.get('fixed_point_rounds_max'), 'host_api', v.get('host_api_frames_per_s'), v.get('host_api_unpipelined_frames_per_s'), 'cpu', v.get('cpu_baseline') and round(v['cpu_baseline']['value']))
    else: print('  f4', k, str(v)[:200])
P
for w in jpeg_decode_b64 jpeg_encode_b64 jpeg_bev_jpeg_b64; do
  timeout 300 python bench.py --workload $w --no-cpu-baseline 2>$O/bench_$w.err | tail -1 > $O/bench_$w.json
  python -c "import json;d=json.load(open('$O/bench_$w.json'));c=d['config'];print('$w',round(d['value']),d['unit'],'ms',round(d['ms_per_step'],3),'rounds',c.get('fixed_point_rounds_max'),'host_api',c.get('host_api_frames_per_s'),c.get('host_api_unpipelined_frames_per_s'),'stage_ms',c.get('host_stage_ms_per_batch'))"
done
timeout 300 python bench.py --workload jpeg_decode_b64 --jpeg-source repo --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_jpeg_decode_b64_repo.json
python -c "import json;d=json.load(open('$O/bench_jpeg_decode_b64_repo.json'));c=d['config'];print('decode repo',round(d['value']),'ms',round(d['ms_per_step'],3),'rounds',c.get('fixed_point_rounds_max'))"
for w in blend_balance_b256 blend_b256; do
  timeout 300 python bench.py --workload $w --placements 3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_$w.json
  python -c "import json;d=json.load(open('$O/bench_$w.json'));o=d.get('other_output_layout');print('$w',round(d['value']),'ms',round(d['ms_per_step'],4),'frac',round(d['roofline']['frac'],3),d['placements']['ms_per_step'],'| other',o and (o['output_layout'],round(o['ms_per_step'],4)))"
done
cd /tmp && export TMPDIR=/tmp
for w in jpeg_decode_b64 jpeg_encode_b64 jpeg_bev_jpeg_b64; do
  rm -rf /tmp/kt_$w /tmp/pv_$w
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$w -- python $R/bench.py --workload $w --steps 6 --warmup 2 --no-cpu-baseline > /tmp/kt_$w.log 2>&1
  cp $(find /tmp/kt_$w -name "*kernel_stats.csv" | head -1) $O/rocprofv3_kernel_stats_$w.csv
  BEVW_BENCH_NO_HOST_API=1 timeout 200 rocprofv3 --pmc SQ_INSTS_VALU --output-format csv -d /tmp/pv_$w -- python $R/bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline > /tmp/pv_$w.log 2>&1
  cp $(find /tmp/pv_$w -name "*counter_collection.csv" | head -1) $O/pmc_valu_$w.csv
done
rm -rf /tmp/kt_rep /tmp/pv_rep
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_rep -- python $R/bench.py --workload jpeg_decode_b64 --jpeg-source repo --steps 6 --warmup 2 --no-cpu-baseline > /tmp/kt_rep.log 2>&1
cp $(find /tmp/kt_rep -name "*kernel_stats.csv" | head -1) $O/rocprofv3_kernel_stats_jpeg_decode_b64_repo.csv
timeout 200 rocprofv3 --pmc SQ_INSTS_VALU --output-format csv -d /tmp/pv_rep -- python $R/bench.py --workload jpeg_decode_b64 --jpeg-source repo --steps 3 --warmup 1 --no-cpu-baseline > /tmp/pv_rep.log 2>&1
cp $(find /tmp/pv_rep -name "*counter_collection.csv" | head -1) $O/pmc_valu_jpeg_decode_b64_repo.csv
cd $R
python tools/r04/jpeg_valu.py $O
