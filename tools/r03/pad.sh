#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | head -3
bash tools/r03/ab.sh padding "direct_stitch_b256 blend_b256 blend_balance_b256 blend_4k" 3 20 nopad:BEVW_UNIT_OWN_PADDING=0 pad:
