#!/bin/bash
# One gpurun call: the decode workloads on several builds of the library (build_var/libbevwarp_<tag>.so, `BEVW_BUILD_TAG=<tag> BEVW_CFLAGS=... python -m
# cameracalibration_amd.build`); "base" = the shipped library.  Per build: kernel averages (rocprofv3) and the decode line.
#   gpurun --timeout 1200 -- 'bash tools/r04/jpeg_variants.sh NAME base nostore ...'
R=$(pwd); N=$1; shift
O=$R/gpurun_out/r04_jv_$N; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for t in "$@"; do
  if [ $t = base ]; then unset BEVW_LIB_PATH; else export BEVW_LIB_PATH=$R/build_var/libbevwarp_$t.so; fi
  for src in ${JV_SRC:-synthetic repo}; do
    rm -rf /tmp/kt
    timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $R/bench.py --workload jpeg_decode_b64 --jpeg-source $src --steps 6 --warmup 2 --no-cpu-baseline > /tmp/kt.log 2>&1
    cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $O/kernel_stats_${t}_$src.csv
    python - $O/kernel_stats_${t}_$src.csv $t $src <<'P'
import csv,sys
f,t,src=sys.argv[1:4]
out=[]
for r in csv.DictReader(open(f)):
    n=r['Name'].split('(')[0].replace('void bevw::jpg::','').replace('bevw::jpg::','')
    if n.startswith('k_jpeg') and float(r['Percentage'])>3.0: out.append(f"{n[7:]} {float(r['AverageNs'])/1e3:.0f}")
print(f"[{t} {src}] "+' | '.join(out))
P
    ( cd $R; timeout 300 python bench.py --workload jpeg_decode_b64 --jpeg-source $src --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys;d=json.loads(sys.stdin.read());print('[$t $src] decode',round(d['value']),'files/s  ms',round(d['ms_per_step'],3))" )
  done
done 2>&1 | tee $O/variants.log
