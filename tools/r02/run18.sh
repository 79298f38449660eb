#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02_run18
mkdir -p $O
cd $R
B="python bench.py --steps 40 --warmup 5 --no-cpu-baseline --workload"
res() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('RESULT $1 ms %.4f median %.4f' % (d['roofline']['kernel_ms'], d['roofline']['kernel_ms_median']))"; }
for rep in 1 2; do
BEVW_PLAN_ONELAUNCH=0 timeout 300 $B direct_stitch_b256 2>&1 | tail -1 | res "per-class product" | tee -a $O/ab.log
BEVW_PLAN_ONELAUNCH=0 BEVW_LIB_PATH=$R/build_abl/libbevwarp_spf.so timeout 300 $B direct_stitch_b256 2>&1 | tail -1 | res "per-class scalar-prefetch" | tee -a $O/ab.log
done
BEVW_PLAN_ONELAUNCH=0 BEVW_LIB_PATH=$R/build_abl/libbevwarp_spf.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "repo_data or full_size or batch_256" 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
for v in product spf; do
L=$R/cameracalibration_amd/libbevwarp.so; [ $v = spf ] && L=$R/build_abl/libbevwarp_spf.so
rm -rf /tmp/kt1; BEVW_LIB_PATH=$L BEVW_PLAN_ONELAUNCH=0 timeout 180 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt1 -- python $R/bench.py --workload direct_stitch_b256 --steps 10 --warmup 2 --no-cpu-baseline > /tmp/kt1.log 2>&1
cp $(find /tmp/kt1 -name "*kernel_stats.csv" | head -1) $O/kernel_stats_per_class_$v.csv; head -7 $O/kernel_stats_per_class_$v.csv | cut -c1-140
done
