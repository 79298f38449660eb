#!/bin/bash
# Round-end evidence, run on the GPU box from the repo root:
#   git rev-parse HEAD > tools/scratch/git_head.txt; git status --porcelain | wc -l >> tools/scratch/git_head.txt; gpurun -- 'bash tools/collect_profiles.sh'
# (.git does not travel to the box: the hash of the tree the collection ran on goes along as a file and is copied beside the results).
# Everything lands under gpurun_out/final/; copy what should be judged into profiles/rNN_final/ (ONE such directory per round).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/final
mkdir -p $O
cd $R
{ echo "collected $(date -u +%Y-%m-%dT%H:%M:%SZ) on $(python -c 'from cameracalibration_amd import _ffi; print(_ffi.device_name(0))' 2>/dev/null)"; echo "git HEAD + number of uncommitted files:"; cat tools/scratch/git_head.txt 2>/dev/null || echo unknown; echo "libbevwarp.so sha256: $(sha256sum cameracalibration_amd/libbevwarp.so | cut -c1-16)"; } > $O/COLLECTION.txt
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; grep -n "passed\|failed" $O/pytest_gpu.log | tail -1
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
# the multi-rank branches of the native RCCL exchange (real librccl with one GPU per rank; the stand-in of tests/native/rccl_standin.cpp on a 1-GPU box)
timeout 600 python -m pytest tests/test_camera_shard.py -m gpu -q -s -k "rccl_world_n_parity" > $O/pytest_rccl_world_n.log 2>&1; grep -E "stand-in|passed|failed|skipped" $O/pytest_rccl_world_n.log | tail -4
cd /tmp && export TMPDIR=/tmp
# row f4: kernel statistics and the VALU instruction counts behind roofline.bound = valu_issue of the JPEG lines
for w in jpeg_decode_b64 jpeg_encode_b64 jpeg_bev_jpeg_b64; do
  rm -rf /tmp/kt_$w /tmp/pv_$w
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$w -- python $R/bench.py --workload $w --steps 6 --warmup 2 --no-cpu-baseline --no-live-traffic > /tmp/kt_$w.log 2>&1
  cp $(find /tmp/kt_$w -name "*kernel_stats.csv" | head -1) $O/rocprofv3_kernel_stats_$w.csv
  BEVW_BENCH_NO_HOST_API=1 timeout 200 rocprofv3 --pmc SQ_INSTS_VALU --output-format csv -d /tmp/pv_$w -- python $R/bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-live-traffic > /tmp/pv_$w.log 2>&1
  cp $(find /tmp/pv_$w -name "*counter_collection.csv" | head -1) $O/pmc_valu_$w.csv
done
rm -rf /tmp/kt_rep /tmp/pv_rep
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_rep -- python $R/bench.py --workload jpeg_decode_b64 --jpeg-source repo --steps 6 --warmup 2 --no-cpu-baseline --no-live-traffic > /tmp/kt_rep.log 2>&1
cp $(find /tmp/kt_rep -name "*kernel_stats.csv" | head -1) $O/rocprofv3_kernel_stats_jpeg_decode_b64_repo_files.csv
timeout 200 rocprofv3 --pmc SQ_INSTS_VALU --output-format csv -d /tmp/pv_rep -- python $R/bench.py --workload jpeg_decode_b64 --jpeg-source repo --steps 3 --warmup 1 --no-cpu-baseline --no-live-traffic > /tmp/pv_rep.log 2>&1
cp $(find /tmp/pv_rep -name "*counter_collection.csv" | head -1) $O/pmc_valu_jpeg_decode_b64_repo.csv
cd $R
python tools/jpeg_valu.py $O | grep "wave-level"
python tools/valu_mix.py $O      # the instruction-mix-weighted issue peak of the JPEG lines (peak_ginst, peak_basis)
python - <<'P'
import json
d=json.load(open('gpurun_out/final/jpeg_valu.json'))
for k,v in d.items():
    if isinstance(v,dict): v.pop('per_kernel_per_step',None)
json.dump(d,open('profiles/jpeg_valu.json','w'),indent=1)   # (on the box: the JPEG lines below price their VALU roofline with THIS build's instruction counts)
P
# the driver's own command: the default line with its f4 summary
( time timeout 900 python bench.py ) > $O/bench_default.json 2> $O/bench_default.time; tail -3 $O/bench_default.time | head -1
for w in direct_stitch_b256 blend_b256 blend_balance_b256 undistort_b64 blend_4k direct_stitch_analytic_f32_b64 direct_stitch_analytic_f64_b64 direct_stitch_analytic_perpixel_b64; do
  timeout 600 python bench.py --workload $w --no-f4 2>/dev/null | tail -1 > $O/bench_$w.json
  python -c "import json;d=json.load(open('$O/bench_$w.json'));o=d.get('other_output_layout');print('$w',round(d['value']),d['unit'],'ms',round(d['ms_per_step'],4),'frac',round(d['roofline']['frac'],3),'placements',d['placements']['ms_per_step'],'| other layout',o and (o['output_layout'],round(o['ms_per_step'],4)),'| cpu',d['cpu_baseline'] and round(d['cpu_baseline']['value'],1))"
done
for w in jpeg_decode_b64 jpeg_encode_b64 jpeg_bev_jpeg_b64; do
  timeout 600 python bench.py --workload $w 2>/dev/null | tail -1 > $O/bench_$w.json
  python -c "import json;d=json.load(open('$O/bench_$w.json'));c=d['config'];print('$w',round(d['value']),d['unit'],'ms',round(d['ms_per_step'],3),'frac',round(d['roofline']['frac'],3),d['roofline']['bound'],'rounds',c.get('fixed_point_rounds_max'),'host_api',c.get('host_api_frames_per_s'),c.get('host_api_unpipelined_frames_per_s'),'| cpu',d['cpu_baseline'] and round(d['cpu_baseline']['value'],1))"
done
timeout 600 python bench.py --workload jpeg_decode_b64 --jpeg-source repo 2>/dev/null | tail -1 > $O/bench_jpeg_decode_b64_repo_files.json
python -c "import json;d=json.load(open('$O/bench_jpeg_decode_b64_repo_files.json'));c=d['config'];print('jpeg_decode repo files',round(d['value']),'ms',round(d['ms_per_step'],3),'rounds',c.get('fixed_point_rounds_max'))"
BEVW_BENCH_BACKEND=gloo timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
  bench.py --gpus 2 --steps 5 --warmup 2 --batch 64 --placements 1 --single-layout 2>/dev/null | tail -1 > $O/bench_two_ranks_one_gpu_gloo.json
python -c "import json;d=json.load(open('$O/bench_two_ranks_one_gpu_gloo.json'));print('2 ranks sharing one GPU (plumbing check):',d['n_gpus'],round(d['value']))"
# the camera-per-GPU bench workload with 2 and 4 ranks on ONE GPU: the bench's pre-timing parity check and its timed pipeline over the library's own
# RCCL layer, the nccl* entry points from the stand-in (rates are those of unix sockets: a plumbing check, never a figure)
if [ "$(python -c 'from cameracalibration_amd import _ffi; print(_ffi.device_count())')" = "1" ]; then
  /opt/rocm/bin/hipcc -O2 -std=c++17 -fPIC -shared tests/native/rccl_standin.cpp -o /tmp/librccl_standin.so -lpthread
  for n in 2 4; do
    BEVW_RCCL_LIB=/tmp/librccl_standin.so BEVW_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2956$n \
      bench.py --gpus $n --workload blend_4k_camera_shard --steps 5 --warmup 2 --batch 8 2>/dev/null | tail -1 > $O/bench_camera_shard_${n}_ranks_one_gpu_standin.json
    python -c "import json;d=json.load(open('$O/bench_camera_shard_${n}_ranks_one_gpu_standin.json'));print('camera shard, $n ranks on one GPU over the RCCL stand-in: parity', d['config'].get('parity_check','')[:6], '| transport', d['config'].get('transport','')[:12])"
  done
fi
cd /tmp && export TMPDIR=/tmp
# kernel statistics: the average over THREE buffer placements (placement variance is +-5 %: one draw can flatter or slander the kernel)
for w in direct_stitch_b256 blend_balance_b256 undistort_b64 blend_b256 blend_4k direct_stitch_analytic_perpixel_b64; do
  rm -rf /tmp/kt_$w
  timeout 180 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$w -- python $R/bench.py --workload $w --steps 10 --warmup 2 --placements 3 --single-layout --no-cpu-baseline --no-live-traffic > /tmp/kt_$w.log 2>&1
  cp $(find /tmp/kt_$w -name "*kernel_stats.csv" | head -1) $O/rocprofv3_kernel_stats_$w.csv
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_${w}_$c
    timeout 120 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_${w}_$c -- python $R/bench.py --workload $w --steps 3 --warmup 1 --placements 1 --single-layout --no-cpu-baseline --no-live-traffic > /tmp/pmc_${w}_$c.log 2>&1
    cp $(find /tmp/pmc_${w}_$c -name "*counter_collection.csv" | head -1) $O/pmc_${w}_$c.csv 2>/dev/null
  done
done
cd $R
python tools/summarize_pmc.py $O
# request / latency counters of the stitch kernel (config 3, both device-image layouts) and of config 4's kernels
bash tools/pmc_merged.sh final/pmc_direct_stitch_aligned direct_stitch_b256 "" > /dev/null 2>&1
bash tools/pmc_merged.sh final/pmc_direct_stitch_dense direct_stitch_b256 "" --output-pitch dense > /dev/null 2>&1
bash tools/pmc_merged.sh final/pmc_blend_balance blend_balance_b256 "" > /dev/null 2>&1
for t in pmc_direct_stitch_aligned pmc_direct_stitch_dense pmc_blend_balance; do cp $O/$t/summary.txt $O/$t.txt; rm -rf $O/$t; done
tail -30 $O/pmc_direct_stitch_aligned.txt
# block timeline of the config-3 and blend steps (tools/block_timeline.py; needs build_var/libbevwarp_x4.so = -DBEVW_EXPERIMENT=4 of the same tree)
if [ -f build_var/libbevwarp_x4.so ]; then
  for v in "" "--blend"; do echo "=== block_timeline $v"; BEVW_LIB_PATH=build_var/libbevwarp_x4.so timeout 300 python tools/block_timeline.py $v; done > $O/block_timeline.log 2>&1
fi
