#!/bin/bash
# Calibrate rocprofv3's FETCH_SIZE / WRITE_SIZE on KNOWN byte counts in the stitch kernels' own access shapes (VERDICT r01
# item 3):  gpurun -- 'bash tools/calibrate_pmc.sh'   ->  gpurun_out/pmc_calibration/{calibration.md, calibration.json}
# tools/hbm_stream --calib launches every kernel once and prints its exact read / write bytes; the two counters are
# collected in separate passes (FETCH_SIZE costs 3 of the 4 TCC slots, WRITE_SIZE 2).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc_calibration
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
$R/tools/hbm_stream --calib > $O/known_bytes.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/cal_$c
  timeout 300 rocprofv3 --pmc $c --output-format csv -d /tmp/cal_$c -- $R/tools/hbm_stream --calib > /tmp/cal_$c.log 2>&1
  cp $(find /tmp/cal_$c -name "*counter_collection.csv" | head -1) $O/$c.csv || { echo "$c pass failed"; tail -3 /tmp/cal_$c.log; }
done
cd $R
python tools/summarize_pmc.py --calibration $O
