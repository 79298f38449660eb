#!/bin/bash
# round 2, GPU call 2: where does the pair-staged step spend its time?  per-class kernel trace + ablation builds
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02_run2
mkdir -p $O
cd $R
B="python bench.py --workload direct_stitch_b256 --steps 40 --warmup 5 --no-cpu-baseline"
res() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('RESULT $1 ms %.4f' % d['roofline']['kernel_ms'])"; }
echo "== product"; timeout 300 $B 2>&1 | tail -1 | res product | tee -a $O/abl.log
for n in 1 2 3 4 5 6; do
  BEVW_LIB_PATH=$R/build_abl/libbevwarp_abl$n.so timeout 300 $B 2>&1 | tail -1 | res abl$n | tee -a $O/abl.log
done
echo "== product again"; timeout 300 $B 2>&1 | tail -1 | res product | tee -a $O/abl.log
for nb in 4 16 32; do BEVW_PLAN_NB=$nb timeout 300 $B 2>&1 | tail -1 | res nb$nb | tee -a $O/abl.log; done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt1; BEVW_PLAN_ONELAUNCH=0 timeout 180 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt1 -- python $R/bench.py --workload direct_stitch_b256 --steps 10 --warmup 2 --no-cpu-baseline > /tmp/kt1.log 2>&1
cp $(find /tmp/kt1 -name "*kernel_stats.csv" | head -1) $O/kernel_stats_per_class.csv; head -8 $O/kernel_stats_per_class.csv | cut -c1-150
rm -rf /tmp/kt2; timeout 180 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt2 -- python $R/bench.py --workload direct_stitch_b256 --steps 10 --warmup 2 --no-cpu-baseline > /tmp/kt2.log 2>&1
cp $(find /tmp/kt2 -name "*kernel_stats.csv" | head -1) $O/kernel_stats_one_launch.csv; head -4 $O/kernel_stats_one_launch.csv | cut -c1-150
