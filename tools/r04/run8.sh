#!/bin/bash
# round 4: JPEG / tools tests after the EXIF fix, then seeded random-case soaks of the stitch (units-only schedule, big class, balance slices), the analytic mode and the codec
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_run8
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_jpeg_gpu.py tests/test_tools.py tests/test_jpeg_goldens.py -m gpu -x -q > $O/pytest_jpeg.log 2>&1; grep -n "passed\|failed" $O/pytest_jpeg.log | tail -1
( time timeout 420 python tools/soak_stitch.py 0 4000 ) > $O/soak_stitch.log 2>&1; tail -4 $O/soak_stitch.log
( time timeout 200 python tools/soak_analytic.py 0 1500 ) > $O/soak_analytic.log 2>&1; tail -4 $O/soak_analytic.log
( time timeout 300 python tools/soak_jpeg.py --cases 3000 --seed 7 ) > $O/soak_jpeg.log 2>&1; tail -4 $O/soak_jpeg.log
