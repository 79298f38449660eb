"""Worker for the camera-per-GPU tests: one process per rank (torch.distributed.run, gloo).

argv: out_dir engine(oracle|hip) blend balance.  Every rank derives the same seeded frame sets, keeps only its own
cameras, runs CameraShardedBev twice (two different stitch ranks) and the stitch rank writes the BEV images.
The exchange goes through tests/_shard_common.GlooTransport -- the CPU stand-in for the RCCL data plane of the product
(a 1-GPU box cannot host a multi-rank RCCL communicator)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch.distributed as dist  # noqa: E402

import _shard_common as SC  # noqa: E402


def main():
    out_dir, engine, blend, balance = sys.argv[1], sys.argv[2], sys.argv[3] == "1", sys.argv[4] == "1"
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    from cameracalibration_amd.SurroundBirdEyeView import cameraShard as CS

    SC.apply_cfg()
    factory = None
    if engine == "oracle":
        from _shard_oracle import OracleShardEngine as factory
    gen = CS.CameraShardedBev(blend, balance, rig=SC.rig(), rank=rank, world_size=world, engine_factory=factory,
                              transport=SC.GlooTransport(rank, world))
    assert gen.world_size == world and gen.rank == rank
    frames, car = SC.frames(batch=2), SC.car()
    mine = np.ascontiguousarray(frames[:, list(gen.cams)])
    for rnd in range(2):
        root = gen.next_root()
        out = gen(mine, car if rnd == 0 else None, root=root)
        assert (out is not None) == (rank == root)
        if out is not None:
            np.save(os.path.join(out_dir, "group%d_round%d.npy" % (gen.group, rnd)), out)
    if engine == "hip":
        # the device-resident pipeline (what bench.py times) must produce the same bytes
        from cameracalibration_amd import _ffi

        pipe = CS.ResidentShardPipeline(gen, mine.shape[0])
        d_frames = _ffi.DeviceBuffer(mine.nbytes).upload(mine)
        d_car = _ffi.DeviceBuffer(car.nbytes).upload(car)
        for rnd in (2, 3):
            root = pipe.step(d_frames.ptr, d_car.ptr)
            if rank == root:
                gen.engine.sync()
                np.save(os.path.join(out_dir, "group%d_round%d.npy" % (gen.group, rnd)),
                        pipe.out.download((mine.shape[0], SC.CFG["BEV_HEIGHT"], SC.CFG["BEV_WIDTH"], 3)))
        pipe.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
