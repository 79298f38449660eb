#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02_run17
mkdir -p $O
cd $R
B="python bench.py --steps 40 --warmup 5 --no-cpu-baseline --workload"
res() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('RESULT $1 ms %.4f median %.4f' % (d['roofline']['kernel_ms'], d['roofline']['kernel_ms_median']))"; }
for rep in 1 2; do
timeout 300 $B direct_stitch_b256 2>&1 | tail -1 | res "product" | tee -a $O/abl.log
for n in 7 9 10 2 1; do
  BEVW_LIB_PATH=$R/build_abl/libbevwarp_abl$n.so timeout 300 $B direct_stitch_b256 2>&1 | tail -1 | res abl$n | tee -a $O/abl.log
done
done
