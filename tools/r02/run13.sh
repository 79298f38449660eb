#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02_run13
mkdir -p $O
cd $R
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --workload"
res() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('RESULT $1 ms %.4f frac %.4f' % (d['roofline']['kernel_ms'], d['roofline']['frac']))"; }
for rep in 1 2; do
for sub in 0 16 32 64 128; do
  BEVW_BAL_SUB=$sub timeout 300 $B blend_balance_b256 2>&1 | tail -1 | res "blend_balance sub$sub" | tee -a $O/ab.log
done
done
BEVW_BAL_SUB=32 timeout 600 python -m pytest tests -m gpu -x -q -k "balance or repo_data or full_size" 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
for sub in 0 32; do
rm -rf /tmp/kt1; BEVW_BAL_SUB=$sub timeout 180 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt1 -- python $R/bench.py --workload blend_balance_b256 --steps 10 --warmup 2 --no-cpu-baseline > /tmp/kt1.log 2>&1
cp $(find /tmp/kt1 -name "*kernel_stats.csv" | head -1) $O/kernel_stats_balance_sub$sub.csv; head -8 $O/kernel_stats_balance_sub$sub.csv | cut -c1-110
done
