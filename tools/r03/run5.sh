#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_run5
mkdir -p $O
cd $R
timeout 600 python tools/r03/placement.py --mode arena --trials 40 --steps 10 > $O/placement_arena.log 2>&1; cat $O/placement_arena.log
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pc_1
timeout 120 rocprofv3 --pmc TCC_EA0_RDREQ TCC_EA0_WRREQ TCC_REQ --kernel-trace --output-format json -d /tmp/pc_1 -- python $R/bench.py --workload direct_stitch_b256 --steps 2 --warmup 1 --no-cpu-baseline > /tmp/pc_1.log 2>&1
f=$(find /tmp/pc_1 -name "*results.json" | head -1)
python - $f $O/sample_results.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
r=d["rocprofiler-sdk-tool"][0]
def trunc(o,depth=0):
    if isinstance(o,dict): return {k:trunc(v,depth+1) for k,v in o.items()}
    if isinstance(o,list): return [trunc(v,depth+1) for v in o[:3]]+(["... %d items"%len(o)] if len(o)>3 else [])
    return o
json.dump(trunc(r),open(sys.argv[2],"w"),indent=1)
PY
ls -la $O
