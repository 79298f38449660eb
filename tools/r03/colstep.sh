#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
bash tools/r03/write_counters.sh nopad:BEVW_UNIT_OWN_PADDING=0 pad: col64:BEVW_UNIT_SKEW=64
cd $R
bash tools/r03/ab.sh colstep "direct_stitch_b256 blend_b256" 3 20 base: col64:BEVW_UNIT_SKEW=64
