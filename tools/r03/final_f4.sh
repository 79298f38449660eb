#!/bin/bash
# Evidence of the tree after row f4 landed, run on the GPU box from the repo root:  gpurun -- 'bash tools/r03/final_f4.sh'
# Full GPU test suite, smoke(), the default bench line (config 3, unchanged kernels) and the JPEG workloads with their CPU baselines and
# rocprofv3 kernel stats.  Lands under gpurun_out/final_f4/ (+ gpurun_out/jpeg/).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/final_f4
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -2 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 600 python bench.py 2>/dev/null | tail -1 > $O/bench_direct_stitch_b256.json
python -c "import json;d=json.load(open('$O/bench_direct_stitch_b256.json'));print('direct_stitch_b256',round(d['value']),'ms',round(d['ms_per_step'],4),'frac',round(d['roofline']['frac'],3),d['placements']['ms_per_step'])"
bash tools/r03/jpeg_profiles.sh
cp $R/gpurun_out/jpeg/* $O/ 2>/dev/null
bash tools/r03/pmc_jpeg.sh > /dev/null 2>&1
cp $R/gpurun_out/pmc_jpeg/decode_b64.txt $O/pmc_jpeg_decode.txt; cp $R/gpurun_out/pmc_jpeg/encode_b64.txt $O/pmc_jpeg_encode.txt
timeout 600 python tools/soak_jpeg.py --cases ${SOAK_CASES:-2500} --seed 2 2>/dev/null | grep "soak_jpeg:" | tee $O/soak_jpeg_seed2.log
