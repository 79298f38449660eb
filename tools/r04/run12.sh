#!/bin/bash
# round 4, last collection of the JPEG evidence on the final build: bench lines, kernel statistics, VALU instruction counts; whole GPU suite; the default line
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_run12
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; grep -n "passed\|failed" $O/pytest_gpu.log | tail -1
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
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
cp $(find /tmp/kt_rep -name "*kernel_stats.csv" | head -1) $O/rocprofv3_kernel_stats_jpeg_decode_b64_repo_files.csv
timeout 200 rocprofv3 --pmc SQ_INSTS_VALU --output-format csv -d /tmp/pv_rep -- python $R/bench.py --workload jpeg_decode_b64 --jpeg-source repo --steps 3 --warmup 1 --no-cpu-baseline > /tmp/pv_rep.log 2>&1
cp $(find /tmp/pv_rep -name "*counter_collection.csv" | head -1) $O/pmc_valu_jpeg_decode_b64_repo.csv
cd $R
python tools/r04/jpeg_valu.py $O | grep "wave-level"
python - <<'P'
import json
d=json.load(open('gpurun_out/r04_run12/jpeg_valu.json'))
for k,v in d.items():
    if isinstance(v,dict): v.pop('per_kernel_per_step',None)
json.dump(d,open('profiles/jpeg_valu.json','w'),indent=1)   # (on the box: the bench lines below read it)
P
for w in jpeg_decode_b64 jpeg_encode_b64 jpeg_bev_jpeg_b64; do
  timeout 600 python bench.py --workload $w 2>/dev/null | tail -1 > $O/bench_$w.json
  python -c "import json;d=json.load(open('$O/bench_$w.json'));c=d['config'];print('$w',round(d['value']),d['unit'],'ms',round(d['ms_per_step'],3),'frac',round(d['roofline']['frac'],3),d['roofline']['bound'],'host_api',c.get('host_api_frames_per_s'),c.get('host_api_unpipelined_frames_per_s'),'| cpu',d['cpu_baseline'] and round(d['cpu_baseline']['value'],1))"
done
timeout 600 python bench.py --workload jpeg_decode_b64 --jpeg-source repo 2>/dev/null | tail -1 > $O/bench_jpeg_decode_b64_repo_files.json
python -c "import json;d=json.load(open('$O/bench_jpeg_decode_b64_repo_files.json'));print('decode repo',round(d['value']),'ms',round(d['ms_per_step'],3),'frac',round(d['roofline']['frac'],3),d['roofline']['bound'])"
( time timeout 900 python bench.py ) > $O/bench_default.json 2> $O/bench_default.time; tail -3 $O/bench_default.time | head -1
python -c "
import json
d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1])
print('default', round(d['value']), 'ms', round(d['ms_per_step'],4), 'frac', round(d['roofline']['frac'],3), 'other', round(d['other_output_layout']['ms_per_step'],4), {k:(round(v['value']), round(v['roofline']['frac'],3)) for k,v in d['f4'].items() if isinstance(v,dict) and 'value' in v})"
