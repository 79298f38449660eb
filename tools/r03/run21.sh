#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_run21
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "output_pitch" 2>&1 | tail -15
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed" | tail -3
