"""Camera-per-GPU mode (cameraShard.CameraShardedBev).

CPU tests run the exchange logic over gloo with the oracle-built stand-in engine (tests/_shard_oracle.py); GPU tests run the
real HIP engine -- single process over every camera partition, and one process per rank (all on cuda:0, gloo transport)
-- and compare BIT-EXACT with the monolithic oracle generator (surroundBEV.py:312-325 restated)."""
import os
import subprocess
import sys

import numpy as np
import pytest

import _shard_common as SC
from conftest import ROOT
from test_dist_cpu import free_port


def reference_bevs(oracle, blend, balance, with_car):
    ref = oracle.RefBevGenerator(SC.rig(), SC.CFG, blend=blend, balance=balance)
    fr, car = SC.frames(batch=2), SC.car()
    return np.stack([ref(*fr[b], car=car if with_car else None) for b in range(fr.shape[0])])


def test_camera_assignment():
    from cameracalibration_amd.SurroundBirdEyeView import cameraShard as CS

    assert CS.camera_assignment(1) == [(0, (0, 1, 2, 3))]
    assert CS.camera_assignment(2) == [(0, (0, 1)), (0, (2, 3))]
    assert CS.camera_assignment(4) == [(0, (0,)), (0, (1,)), (0, (2,)), (0, (3,))]
    a8 = CS.camera_assignment(8)
    assert [g for g, _ in a8] == [0, 0, 0, 0, 1, 1, 1, 1] and [c for _, c in a8] == [(0,), (1,), (2,), (3,)] * 2
    assert CS.group_ranks(8, 1) == [4, 5, 6, 7]
    for bad in (3, 5, 6):
        with pytest.raises(Exception):
            CS.camera_assignment(bad)
    # every camera exactly once per group
    for w in (1, 2, 4, 8, 12):
        for g in {g for g, _ in CS.camera_assignment(w)}:
            cams = sorted(c for gg, cs in CS.camera_assignment(w) if gg == g for c in cs)
            assert cams == [0, 1, 2, 3]


@pytest.mark.parametrize("blend,balance", [(False, False), (True, True)])
def test_stand_in_engine_matches_reference_single_rank(oracle, blend, balance):
    """Pins the checker: the oracle-built shard engine, world 1, equals the monolithic oracle generator."""
    from cameracalibration_amd.SurroundBirdEyeView import cameraShard as CS
    from _shard_oracle import OracleShardEngine

    SC.apply_cfg()
    gen = CS.CameraShardedBev(blend, balance, rig=SC.rig(), rank=0, world_size=1, engine_factory=OracleShardEngine)
    out = gen(SC.frames(batch=2), SC.car())
    assert np.array_equal(out, reference_bevs(oracle, blend, balance, True))


def run_workers(tmp_path, world, engine, blend, balance, env_extra):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
           "127.0.0.1", "--master-port", str(free_port()), os.path.join(ROOT, "tests", "_shard_worker.py"), str(tmp_path), engine,
           "1" if blend else "0", "1" if balance else "0"]
    env = dict(os.environ, OMP_NUM_THREADS="2", **env_extra)
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]


def check_outputs(tmp_path, oracle, world, blend, balance, resident=False):
    ngroups = max(1, world // 4)
    for g in range(ngroups):
        for rnd, with_car in ((0, True), (1, False)) + (((2, True), (3, True)) if resident else ()):
            got = np.load(tmp_path / ("group%d_round%d.npy" % (g, rnd)))
            want = reference_bevs(oracle, blend, balance, with_car)
            assert got.shape == want.shape
            assert np.array_equal(got, want), "group %d round %d: %d bytes differ" % (g, rnd, np.count_nonzero(got != want))


@pytest.mark.parametrize("world,blend,balance", [(2, False, False), (4, True, True)])
def test_gloo_exchange_with_stand_in_engine(tmp_path, oracle, world, blend, balance):
    run_workers(tmp_path, world, "oracle", blend, balance, dict(HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES=""))
    check_outputs(tmp_path, oracle, world, blend, balance)


# ---------------------------------------------------------------------------------------------------------------
# GPU
# ---------------------------------------------------------------------------------------------------------------
PARTITIONS = [[(0, 1, 2, 3)], [(0, 1), (2, 3)], [(0,), (1,), (2,), (3,)], [(0, 2), (1, 3)], [(0,), (1, 2, 3)]]


@pytest.mark.gpu
@pytest.mark.parametrize("blend,balance", [(False, False), (True, False), (False, True), (True, True)])
def test_hip_shards_combine_to_the_full_stitch(oracle, blend, balance):
    from cameracalibration_amd.SurroundBirdEyeView import cameraShard as CS

    SC.apply_cfg()
    CS._sb.BevGenerator.init_args(None)
    rig_list = [SC.rig()[n] for n in SC.W.CAMERA_NAMES]
    frames, car = SC.frames(batch=3), SC.car()
    ref = oracle.RefBevGenerator(SC.rig(), SC.CFG, blend=blend, balance=balance)
    want = np.stack([ref(*frames[b], car=car) for b in range(frames.shape[0])])
    for part in PARTITIONS:
        engines = [CS.HipShardEngine(rig_list, cams, blend, balance) for cams in part]
        all_vsums = None
        if balance:
            all_vsums = np.zeros((frames.shape[0], 4), np.uint64)
            for e in engines:
                all_vsums[:, list(e.cams)] = e.vsums(np.ascontiguousarray(frames[:, list(e.cams)]))
            for b in range(frames.shape[0]):
                assert [int(v) for v in all_vsums[b]] == [oracle.sum_v(frames[b, k]) for k in range(4)]
        parts = [e.partial(np.ascontiguousarray(frames[:, list(e.cams)]), all_vsums) for e in engines]
        boxes = [e.box for e in engines]
        got = engines[-1].combine(parts, boxes, car)
        assert np.array_equal(got, want), "partition %s: %d bytes differ" % (part, np.count_nonzero(got != want))
        for e in engines:
            e.close()


@pytest.mark.gpu
def test_hip_shard_box_and_errors():
    from cameracalibration_amd import _ffi
    from cameracalibration_amd.SurroundBirdEyeView import cameraShard as CS
    from _shard_oracle import OracleShardEngine

    SC.apply_cfg()
    CS._sb.BevGenerator.init_args(None)
    rig_list = [SC.rig()[n] for n in SC.W.CAMERA_NAMES]
    for cams in [(0,), (1,), (2,), (3,), (0, 1), (2, 3)]:
        for blend in (False, True):
            e = CS.HipShardEngine(rig_list, cams, blend, False)
            assert e.box == OracleShardEngine(rig_list, cams, blend, False).box
            if cams == (0,):
                # a shard handle refuses the whole-rig entry points and tables of cameras it does not own
                d = _ffi.DeviceBuffer(16)
                assert _ffi.lib().bevw_run_device(e.h, d.ptr, 1, None, d.ptr) == -1
                m = np.zeros((SC.CFG["BEV_HEIGHT"], SC.CFG["BEV_WIDTH"], 2), np.int16)
                assert _ffi.lib().bevw_get_lut(e.h, 1, _ffi.ptr(m), None) == -1
                assert _ffi.lib().bevw_get_lut(e.h, 0, _ffi.ptr(m), None) == 0
                d.free()
            e.close()
    bad = np.asarray([1, 0], np.int32)
    e = CS.HipShardEngine(rig_list, (0,), False, False)
    assert _ffi.lib().bevw_set_camera_shard(e.h, _ffi.ptr(bad), 2) == -1
    e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("world,blend,balance", [(2, True, True), (4, False, False), (4, True, True)])
def test_one_process_per_camera_on_the_gpu(tmp_path, oracle, world, blend, balance):
    """The real N > 1 path: one process per rank, HIP engine in every rank (all ranks share cuda:0 on the 1-GPU box),
    parts exchanged over torch.distributed (gloo here, RCCL on a multi-GPU node)."""
    run_workers(tmp_path, world, "hip", blend, balance, {})
    check_outputs(tmp_path, oracle, world, blend, balance, resident=True)


@pytest.mark.gpu
@pytest.mark.parametrize("blend,balance", [(False, False), (True, True)])
def test_resident_pipeline_single_rank(oracle, blend, balance):
    from cameracalibration_amd import _ffi
    from cameracalibration_amd.SurroundBirdEyeView import cameraShard as CS

    SC.apply_cfg()
    gen = CS.CameraShardedBev(blend, balance, rig=SC.rig(), rank=0, world_size=1)
    frames, car = SC.frames(batch=2), SC.car()
    pipe = CS.ResidentShardPipeline(gen, 2)
    d_frames, d_car = _ffi.DeviceBuffer(frames.nbytes).upload(frames), _ffi.DeviceBuffer(car.nbytes).upload(car)
    assert pipe.step(d_frames.ptr, d_car.ptr) == 0
    gen.engine.sync()
    got = pipe.out.download((2, SC.CFG["BEV_HEIGHT"], SC.CFG["BEV_WIDTH"], 3))
    assert np.array_equal(got, reference_bevs(oracle, blend, balance, True))
    assert np.array_equal(gen(frames, car), got)


RCCL_VIEW_SCRIPT = """
import torch                      # torch first: libbevwarp then binds to the HIP runtime torch already loaded
import numpy as np
from cameracalibration_amd import _ffi
from cameracalibration_amd.SurroundBirdEyeView import cameraShard as CS
src = np.arange(4096, dtype=np.uint8)
buf = _ffi.DeviceBuffer(src.nbytes).upload(src)
t = torch.as_tensor(CS._CudaView(buf.ptr, buf.nbytes), device="cuda")
assert t.data_ptr() == buf.ptr and t.dtype == torch.uint8 and t.numel() == 4096
assert np.array_equal(t.cpu().numpy(), src)
t.add_(1)
torch.cuda.synchronize()
assert np.array_equal(buf.download((4096,)), src + np.uint8(1))
print("aliased")
"""


@pytest.mark.gpu
def test_device_buffers_are_visible_to_torch_for_rccl():
    """RCCL moves the parts straight out of bevw_malloc memory: the zero-copy torch view must alias it.  Own process,
    torch imported first -- the order bench.py and the workers use (two HIP runtimes must not meet in one process)."""
    out = subprocess.run([sys.executable, "-c", RCCL_VIEW_SCRIPT], capture_output=True, text=True, cwd=ROOT, timeout=300)
    assert out.returncode == 0 and "aliased" in out.stdout, out.stdout[-1000:] + out.stderr[-3000:]
