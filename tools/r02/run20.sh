#!/bin/bash
# cleaned build (sector-staged schedule and ablation hooks removed): GPU suite + one bench line per workload
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02_run20
mkdir -p $O
cd $R
echo "== pytest"; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee $O/pytest.log
B="python bench.py --steps 40 --warmup 5 --no-cpu-baseline --workload"
res() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('RESULT $1 ms %.4f median %.4f frac %.4f' % (d['roofline']['kernel_ms'], d['roofline'].get('kernel_ms_median', 0), d['roofline']['frac']))"; }
for w in direct_stitch_b256 blend_b256 undistort_b64 blend_4k blend_balance_b256; do
  timeout 300 $B $w 2>&1 | tail -1 | tee -a $O/lines.jsonl | res "$w" | tee -a $O/ab.log
done
timeout 300 $B direct_stitch_b256 2>&1 | tail -1 | res "direct_stitch_b256 again" | tee -a $O/ab.log
