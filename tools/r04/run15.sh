#!/bin/bash
# after the walker rewrites: JPEG soak against libjpeg-turbo + the decode / default lines with this build's VALU counts (profiles/jpeg_valu.json)
R=$(pwd); O=$R/gpurun_out/r04_run15; mkdir -p $O
BEVW_SOAK_SECONDS=${1:-150} timeout 900 python tools/soak_jpeg.py --cases 100000 --seed 4 2>&1 | tail -4 | tee $O/soak_jpeg.log
for a in "" "--jpeg-source repo"; do
  n=$( [ -z "$a" ] && echo jpeg_decode_b64 || echo jpeg_decode_b64_repo_files )
  timeout 600 python bench.py --workload jpeg_decode_b64 $a 2>/dev/null | tail -1 > $O/bench_$n.json
  python -c "import json;d=json.load(open('$O/bench_$n.json'));print('$n',round(d['value']),'ms',round(d['ms_per_step'],3),'frac',round(d['roofline']['frac'],3),d['roofline']['bound'],'rounds',d['config'].get('fixed_point_rounds_max'))"
done
( time timeout 900 python bench.py ) > $O/bench_default.json 2> $O/bench_default.time; tail -3 $O/bench_default.time | head -1
python -c "
import json
d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1])
print('default',round(d['value']),'ms',round(d['ms_per_step'],4),'frac',round(d['roofline']['frac'],3),{k:(round(v['value']),round(v['roofline']['frac'],3)) for k,v in d.get('f4',{}).items() if isinstance(v,dict) and 'value' in v})"
