#!/bin/bash
# round 2, GPU call 5: pair schedule v3 (buffer loads with masked lanes, whole-tile staging up to 4 rounds): parity, lane
# permutation of the group loads, tile shape, ablations
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02_run5
mkdir -p $O
cd $R
echo "== pytest (pair v3)"; timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee $O/pytest_pair.log
echo "== pytest perm 2"; BEVW_PAIR_PERM=2 timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee $O/pytest_pair_perm2.log
B="python bench.py --steps 40 --warmup 5 --no-cpu-baseline --workload"
res() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('RESULT $1 ms %.4f frac %.4f tiles %s' % (d['roofline']['kernel_ms'], d['roofline']['frac'], d['config'].get('tiles')))"; }
for perm in 0 1 2; do
  for rep in 1 2; do
    BEVW_PAIR_PERM=$perm timeout 300 $B direct_stitch_b256 2>&1 | tail -1 | res "direct perm$perm one_launch" | tee -a $O/ab.log
  done
  BEVW_PAIR_PERM=$perm BEVW_PLAN_ONELAUNCH=0 timeout 300 $B direct_stitch_b256 2>&1 | tail -1 | res "direct perm$perm per_class" | tee -a $O/ab.log
done
for lx in 4 16; do
  BEVW_PLAN_LX=$lx timeout 300 $B direct_stitch_b256 2>&1 | tail -1 | res "direct lx$lx" | tee -a $O/ab.log
done
for w in blend_b256 undistort_b64 blend_4k; do
  for perm in 0 2; do BEVW_PAIR_PERM=$perm timeout 300 $B $w 2>&1 | tail -1 | res "$w perm$perm" | tee -a $O/ab.log; done
done
for n in 1 2 5; do
  BEVW_LIB_PATH=$R/build_abl/libbevwarp_abl$n.so timeout 300 $B direct_stitch_b256 2>&1 | tail -1 | res abl$n | tee -a $O/abl.log
done
cd /tmp && export TMPDIR=/tmp
for perm in 0 2; do
rm -rf /tmp/kt1; BEVW_PAIR_PERM=$perm BEVW_PLAN_ONELAUNCH=0 timeout 180 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt1 -- python $R/bench.py --workload direct_stitch_b256 --steps 10 --warmup 2 --no-cpu-baseline > /tmp/kt1.log 2>&1
cp $(find /tmp/kt1 -name "*kernel_stats.csv" | head -1) $O/kernel_stats_per_class_perm$perm.csv; head -9 $O/kernel_stats_per_class_perm$perm.csv | cut -c1-150
done
