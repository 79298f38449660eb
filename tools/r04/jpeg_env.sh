#!/bin/bash
# One gpurun call: the decode line under several environments ("label:VAR=..,VAR=..").   gpurun -- 'bash tools/r04/jpeg_env.sh NAME p2: p3s3:BEVW_JPEG_PARTS=3,BEVW_JPEG_STREAMS=3'
R=$(pwd); N=$1; shift
O=$R/gpurun_out/r04_je_$N; mkdir -p $O
for v in "$@"; do
  label=${v%%:*}; envs=${v#*:}
  for src in ${JV_SRC:-synthetic repo}; do
    ( IFS=,; for e in $envs; do [ -n "$e" ] && export "$e"; done; unset IFS
      timeout 300 python bench.py --workload jpeg_decode_b64 --jpeg-source $src --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys;d=json.loads(sys.stdin.read());print('[$label $src] decode',round(d['value']),'files/s  ms',round(d['ms_per_step'],3))" )
  done
done 2>&1 | tee $O/env.log
