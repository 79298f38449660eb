"""Worker of tests/test_camera_shard.py::test_rccl_world_n_parity: ONE PROCESS PER GPU, the product's own transport.

argv: out_dir.  Rank / world / device come from the launcher's environment (RANK, WORLD_SIZE, LOCAL_RANK, MASTER_ADDR, MASTER_PORT -- no
torch anywhere).  Every rank derives the same seeded 4K frame sets, keeps its own cameras, and runs CameraShardedBev(blend, balance) over
libbevwarp's native RCCL layer (csrc/bevw_comm.h: ncclAllGather of the V sums, grouped ncclSend / ncclRecv of the mask boxes) twice through
the host-array call and twice through the device-resident pipeline (four different stitch ranks); stitch ranks write their BEV batches."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

import _shard_common as SC  # noqa: E402
from cameracalibration_amd import _ffi, workloads as W  # noqa: E402

BATCH = 3


def inputs():
    cfg = W.CONFIG_4K
    return cfg, W.rig_4k(), SC.frames(batch=BATCH, seed=91, cfg=cfg), SC.car(cfg)


def main():
    out_dir = sys.argv[1]
    from cameracalibration_amd.SurroundBirdEyeView import cameraShard as CS

    rank, world, local, _, _ = CS.launcher_env()
    cfg, rig, frames, car = inputs()
    SC.apply_cfg(cfg)
    dev = local if _ffi.device_count() > local else 0
    with CS.CameraShardedBev(True, True, rig=rig, rank=rank, world_size=world, device=dev) as gen:
        assert gen.world_size == world and (world == 1 or gen.rccl is not None), "the native RCCL transport must be the one in use"
        mine = np.ascontiguousarray(frames[:, list(gen.cams)])
        for rnd in range(2):
            root = gen.next_root()
            out = gen(mine, car if rnd == 0 else None, root=root)
            assert (out is not None) == (rank == root)
            if out is not None:
                np.save(os.path.join(out_dir, "group%d_round%d.npy" % (gen.group, rnd)), out)
        pipe = CS.ResidentShardPipeline(gen, BATCH)
        d_frames = _ffi.DeviceBuffer(mine.nbytes, dev).upload(mine)
        d_car = _ffi.DeviceBuffer(car.nbytes, dev).upload(car)
        for rnd in (2, 3):
            root = pipe.step(d_frames.ptr, d_car.ptr if rnd == 2 else None)
            if rank == root:
                gen.engine.sync()
                np.save(os.path.join(out_dir, "group%d_round%d.npy" % (gen.group, rnd)),
                        pipe.out.download((BATCH, cfg["BEV_HEIGHT"], cfg["BEV_WIDTH"], 3)))
        pipe.close()
    print("rank %d of %d on device %d done" % (rank, world, dev))


if __name__ == "__main__":
    main()
