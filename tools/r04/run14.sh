#!/bin/bash
# round 4, run 14+: JPEG decode after a codec change -- the JPEG GPU tests, kernel stats of the two decode workloads, the decode lines
R=$(pwd); cd /tmp; export TMPDIR=/tmp
O=$R/gpurun_out/r04_run14
mkdir -p $O
( cd $R && timeout 900 python -m pytest tests -m gpu -x -q -k "jpeg or golden or pipeline" 2>&1 | tail -3 ) | tee $O/pytest_gpu_jpeg.log
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_dec -- python $R/bench.py --workload jpeg_decode_b64 --steps 6 --warmup 2 --no-cpu-baseline > /tmp/kt_dec.log 2>&1
cp $(find /tmp/kt_dec -name "*kernel_stats.csv" | head -1) $O/rocprofv3_kernel_stats_jpeg_decode_b64.csv
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_rep -- python $R/bench.py --workload jpeg_decode_b64 --jpeg-source repo --steps 6 --warmup 2 --no-cpu-baseline > /tmp/kt_rep.log 2>&1
cp $(find /tmp/kt_rep -name "*kernel_stats.csv" | head -1) $O/rocprofv3_kernel_stats_jpeg_decode_b64_repo_files.csv
cd $R
python - <<'P'
import csv
for f in ['rocprofv3_kernel_stats_jpeg_decode_b64.csv','rocprofv3_kernel_stats_jpeg_decode_b64_repo_files.csv']:
    print(f)
    for r in csv.DictReader(open('gpurun_out/r04_run14/'+f)):
        n=r['Name'].split('(')[0].replace('void bevw::jpg::','')
        if float(r['Percentage'])>1.0: print(f"  {n:28s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:8.1f} us")
P
for a in "" "--jpeg-source repo"; do
  timeout 600 python bench.py --workload jpeg_decode_b64 $a --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_decode.json
  python -c "import json;d=json.load(open('$O/bench_decode.json'));print('decode $a',round(d['value']),'ms',round(d['ms_per_step'],3),'frac',round(d['roofline']['frac'],3))"
done
