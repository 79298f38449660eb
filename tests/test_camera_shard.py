"""Camera-per-GPU mode (cameraShard.CameraShardedBev).

CPU tests run the exchange logic over gloo (tests/_shard_common.GlooTransport, the stand-in for the product's native RCCL data
plane) with the oracle-built stand-in engine (tests/_shard_oracle.py); GPU tests run the real HIP engine -- single process over
every camera partition, one process per rank (all on cuda:0, gloo stand-in), and the RCCL layer itself on a 1-rank communicator
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
@pytest.mark.parametrize("bw", [250, 249])
def test_hip_shards_with_a_bev_width_that_is_not_a_multiple_of_4(oracle, bw):
    """camera shards on the padded-pitch tile plan (Plan::pitch): boxes, packing and the combine take the byte-granular paths"""
    from cameracalibration_amd.SurroundBirdEyeView import cameraShard as CS

    cfg = dict(SC.CFG, BEV_WIDTH=bw, BEV_HEIGHT=251)
    SC.apply_cfg(cfg)
    try:
        CS._sb.BevGenerator.init_args(None)
        rig_list = [SC.rig()[n] for n in SC.W.CAMERA_NAMES]
        frames, car = SC.frames(batch=2, cfg=cfg), SC.car(cfg)
        for blend, balance in [(True, False), (True, True)]:
            ref = oracle.RefBevGenerator(SC.rig(), cfg, blend=blend, balance=balance)
            want = np.stack([ref(*frames[b], car=car) for b in range(frames.shape[0])])
            for part in ([(0, 1), (2, 3)], [(0,), (1,), (2,), (3,)]):
                engines = [CS.HipShardEngine(rig_list, cams, blend, balance) for cams in part]
                all_vsums = None
                if balance:
                    all_vsums = np.zeros((frames.shape[0], 4), np.uint64)
                    for e in engines:
                        all_vsums[:, list(e.cams)] = e.vsums(np.ascontiguousarray(frames[:, list(e.cams)]))
                parts = [e.partial(np.ascontiguousarray(frames[:, list(e.cams)]), all_vsums) for e in engines]
                got = engines[-1].combine(parts, [e.box for e in engines], car)
                assert np.array_equal(got, want), "bw %d partition %s: %d bytes differ" % (bw, part, np.count_nonzero(got != want))
                for e in engines:
                    e.close()
    finally:
        SC.apply_cfg()


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
    """The N > 1 control flow with the real engine: one process per rank, HIP engine in every rank (all ranks share cuda:0 on
    the 1-GPU box), parts exchanged through the gloo stand-in transport (tests/_shard_common.py).  The RCCL data plane itself
    needs one GPU per rank; its 1-rank self-test is test_rccl_transport_world_1_self_test."""
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


@pytest.mark.gpu
def test_rccl_transport_world_1_self_test():
    """The native RCCL layer (csrc/bevw_comm.h) on what a 1-GPU box can run: librccl is dlopen'ed, a 1-rank communicator is
    created from a unique id, an all-gather and a grouped send / receive to self run on the engine's own stream and deliver
    the bytes; the V-sum all-gather entry returns the [batch][4] layout."""
    import ctypes as C

    from cameracalibration_amd import _ffi
    from cameracalibration_amd.SurroundBirdEyeView import cameraShard as CS

    L = _ffi.lib()
    assert L.bevw_comm_available() == 1
    SC.apply_cfg()
    gen = CS.CameraShardedBev(True, True, rig=SC.rig(), rank=0, world_size=1)
    e = gen.engine
    ident = (C.c_uint8 * 128)()
    _ffi.check(L.bevw_comm_unique_id(ident))
    assert any(ident)
    comm = C.c_void_p()
    _ffi.check(L.bevw_comm_create(0, 0, 1, ident, C.byref(comm)))
    try:
        src = np.random.default_rng(3).integers(0, 256, 1 << 20, dtype=np.uint8)
        d_src, d_dst = _ffi.DeviceBuffer(src.nbytes).upload(src), _ffi.DeviceBuffer(src.nbytes)
        d_dst.fill(0)
        _ffi.check(L.bevw_comm_selftest(e.h, comm, d_src.ptr, d_dst.ptr, src.nbytes))
        e.sync()
        assert np.array_equal(d_dst.download(src.shape), src)
        # V sums of the one rank that owns all four cameras: gathered == its own, in [batch][4] order
        frames = SC.frames(batch=3)
        d_frames = _ffi.DeviceBuffer(frames.nbytes).upload(frames)
        vs, vs_all = _ffi.DeviceBuffer(8 * 3 * 4), _ffi.DeviceBuffer(8 * 3 * 4)
        e.vsums_device(d_frames.ptr, 3, vs.ptr)
        _ffi.check(L.bevw_shard_allgather_vsums(e.h, comm, vs.ptr, 3, vs_all.ptr))
        e.sync()
        assert np.array_equal(vs_all.download((3, 4), np.uint64), vs.download((3, 4), np.uint64))
        # a 1-rank gather is a no-op and must not touch anything
        assert L.bevw_shard_gather_parts(e.h, comm, d_src.ptr, src.nbytes, 0, None, None) == 0
        for b in (d_src, d_dst, d_frames, vs, vs_all):
            b.free()
    finally:
        L.bevw_comm_destroy(comm)
        gen.close()


def rccl_world_n_parity(out_dir, world, extra_env=None):
    """Spawns one worker per GPU (tests/_rccl_worker.py, the product's native RCCL transport, no torch) and compares what the stitch ranks
    wrote with BevGenerator(blend=True, balance=True) on the same 4K frame sets -- bit-exact.  Also used by bench.py before it times the
    camera-shard workload on more than one GPU."""
    import _rccl_worker as RW
    from cameracalibration_amd.SurroundBirdEyeView import surroundBEV as SB

    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0", **(extra_env or {}))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_rccl_worker.py"), str(out_dir)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = []
    for p in procs:
        try:
            logs.append(p.communicate(timeout=600)[0])
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise AssertionError("the RCCL workers did not finish: " + "\n".join(logs))
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)[-4000:]
    cfg, rig, frames, car = RW.inputs()
    SC.apply_cfg(cfg)
    try:
        bev = SB.BevGenerator(blend=True, balance=True, rig=rig)
        want_car, want = bev.batch(frames, car), bev.batch(frames)
    finally:
        SC.apply_cfg()
    for rnd in range(4):
        got = np.load(os.path.join(str(out_dir), "group0_round%d.npy" % rnd))
        assert np.array_equal(got, want_car if rnd in (0, 2) else want), "round %d of world %d differs from BevGenerator" % (rnd, world)
    return logs


def build_rccl_standin(out_dir):
    """tests/native/rccl_standin.cpp -> <out_dir>/librccl_standin.so (host code only: seconds)."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    lib = os.path.join(str(out_dir), "librccl_standin.so")
    r = subprocess.run([hipcc, "-O2", "-std=c++17", "-fPIC", "-shared", os.path.join(ROOT, "tests", "native", "rccl_standin.cpp"), "-o", lib, "-lpthread"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    return lib


@pytest.mark.gpu
@pytest.mark.parametrize("world", [1, 2, 4])
def test_rccl_world_n_parity(tmp_path, world):
    """The rank > 0 branches of the native RCCL exchange (bevw_shard_gather_parts, bevw_shard_allgather_vsums, k_vsums_interleave): one
    process per rank on the 4K rig with blend + balance, bit-exact against BevGenerator.  With one GPU per rank the library is the real
    librccl; on a box with FEWER GPUs than ranks (real RCCL refuses two ranks on one device) the ranks share device 0 and the nine
    entry points the product dlsym()s come from tests/native/rccl_standin.cpp through BEVW_RCCL_LIB -- every line of the product's
    exchange code above the nccl* calls is the one an 8-GPU node runs.  World 1 runs the same worker without an exchange."""
    from cameracalibration_amd import _ffi

    env = {}
    if _ffi.device_count() < world:
        if _ffi.device_count() < 1:
            pytest.skip("no GPU")
        env["BEVW_RCCL_LIB"] = build_rccl_standin(tmp_path)
        print("world %d on %d GPU(s): the RCCL stand-in carries the exchange" % (world, _ffi.device_count()))
    elif world > 1:
        assert _ffi.lib().bevw_comm_available() == 1
    rccl_world_n_parity(tmp_path, world, env)


@pytest.mark.parametrize("world", [2, 4])
def test_rccl_standin_mesh_on_host_buffers(tmp_path, world):
    """The stand-in itself, without a GPU: its socket mesh, grouped send / receive (boxes far beyond a socket buffer, every rank once the
    root, a ring with sends and receives on every rank, a pair to itself), the all-gather and the size check, on host buffers
    (BEVW_RCCL_STANDIN_HOST=1), one process per rank through ctypes."""
    import ctypes as C

    if not os.path.exists(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")):
        pytest.skip("hipcc not available")
    lib = build_rccl_standin(tmp_path)
    os.environ["BEVW_RCCL_STANDIN_HOST"] = "1"
    try:
        L = C.CDLL(lib)
        ident = (C.c_char * 128)()
        assert L.ncclGetUniqueId(ident) == 0
    finally:
        del os.environ["BEVW_RCCL_STANDIN_HOST"]
    raw = bytes(ident)
    assert b"bevw-rccl-standin" in raw
    env = dict(os.environ, BEVW_RCCL_STANDIN_HOST="1")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_rccl_standin_worker.py"), lib, raw.hex(), str(r), str(world)], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    logs = []
    for p in procs:
        try:
            logs.append(p.communicate(timeout=120)[0])
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise AssertionError("the stand-in workers did not finish: " + "\n".join(logs))
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)[-4000:]
    for r in range(world):
        assert "standin rank %d of %d ok" % (r, world) in logs[r]


def test_control_channel_over_sockets():
    """SocketGroup (the out-of-band channel that carries the RCCL unique id and the mask boxes): 3 ranks as threads."""
    import threading

    from cameracalibration_amd.SurroundBirdEyeView import cameraShard as CS

    port, out = free_port(), {}

    def run(r):
        g = CS.SocketGroup(r, 3, "127.0.0.1", port)
        out[r] = (g.broadcast(b"x" * 128 if r == 0 else b"", 128), g.all_gather(bytes([r]) * 16))
        g.close()

    ts = [threading.Thread(target=run, args=(r,)) for r in range(3)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(30)
    for r in range(3):
        assert out[r][0] == b"x" * 128
        assert out[r][1] == [bytes([k]) * 16 for k in range(3)]


def test_product_package_is_free_of_pytorch():
    """north_star: "host code stays Python calling a thin ctypes C-ABI .so (no PyTorch ...)": nothing under
    cameracalibration_amd/ mentions it; torch.distributed appears only in bench.py (multi-rank barrier) and in tests/."""
    hits = []
    pkg = os.path.join(ROOT, "cameracalibration_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip")):
                if "torch" in open(os.path.join(d, f), errors="replace").read():
                    hits.append(os.path.relpath(os.path.join(d, f), ROOT))
    assert not hits, hits


def test_control_hub_survives_a_silent_connection():
    """a connection that never sends its rank hello is dropped after a short per-connection deadline; the hub keeps accepting
    and the group still forms."""
    import socket
    import threading
    import time

    from cameracalibration_amd.SurroundBirdEyeView import cameraShard as CS

    port, out = free_port(), {}

    def run(r, delay):
        time.sleep(delay)
        g = CS.SocketGroup(r, 2, "127.0.0.1", port, timeout=20.0)
        out[r] = g.all_gather(bytes([r]) * 4)
        g.close()

    hub = threading.Thread(target=run, args=(0, 0.0))
    hub.start()
    time.sleep(0.3)
    stray = socket.create_connection(("127.0.0.1", port), timeout=5.0)   # connects first, says nothing
    member = threading.Thread(target=run, args=(1, 0.2))
    member.start()
    t0 = time.time()
    hub.join(15)
    member.join(15)
    stray.close()
    assert not hub.is_alive() and not member.is_alive() and time.time() - t0 < 10
    assert out[0] == out[1] == [b"\x00" * 4, b"\x01" * 4]
