#!/bin/bash
# ablation with the spatial unit order: every output row on a 64-byte boundary (pitch 1088 pixels, unit columns at multiples of 64 pixels)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_run20
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
for v in base minw64 aligned aligned_minw16; do
  case $v in base) E="";; minw64) E="BEVW_UNIT_MIN_W=64";; aligned) E="BEVW_UNIT_MIN_W=64 BEVW_ABL_PITCH_ALIGN=64";; aligned_minw16) E="BEVW_ABL_PITCH_ALIGN=64";; esac
  rm -rf /tmp/kt_$v
  env $E timeout 180 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$v -- python $R/bench.py --workload direct_stitch_b256 --steps 10 --warmup 2 --placements 3 --no-cpu-baseline > /tmp/kt_$v.log 2>&1
  f=$(find /tmp/kt_$v -name "*kernel_stats.csv" | head -1)
  echo "$v rep $rep: $(grep k_plan_all $f | cut -d, -f1-5)"
done
done 2>&1 | tee $O/ablation.log
