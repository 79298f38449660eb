#!/bin/bash
# the camera-shard bench workload with 2 and 4 ranks on ONE GPU: torch.distributed (gloo) for the bench's barrier, the library's own RCCL layer over the stand-in
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_call16
mkdir -p $O
cd $R
/opt/rocm/bin/hipcc -O2 -std=c++17 -fPIC -shared tests/native/rccl_standin.cpp -o /tmp/librccl_standin.so -lpthread
for n in 2 4; do
  BEVW_RCCL_LIB=/tmp/librccl_standin.so BEVW_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2956$n \
    bench.py --gpus $n --workload blend_4k_camera_shard --steps 5 --warmup 2 --batch 8 > $O/bench_camera_shard_${n}_ranks_one_gpu.log 2>&1
  tail -1 $O/bench_camera_shard_${n}_ranks_one_gpu.log | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); c=d['config']; print('$n ranks on one GPU:', d['n_gpus'], 'gpu', d['ranks'], 'ranks', round(d['value']), d['unit'], 'ms', round(d['ms_per_step'],3), '| parity:', c.get('parity_check'), '| transport:', c.get('transport'))
except Exception as e: print('FAILED', e)"
done
tail -5 $O/bench_camera_shard_2_ranks_one_gpu.log | cut -c1-300
