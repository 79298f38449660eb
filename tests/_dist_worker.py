"""Worker for tests/test_dist_cpu.py: launched by torch.distributed.run with world_size 2 on CPU (gloo)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


def main():
    out_dir = sys.argv[1]
    d = bench.Dist()
    assert d.world == 2 and d.rank in (0, 1)
    d.barrier()
    # max-over-ranks timing and whole-job aggregation exactly as bench.main() does them
    wall = d.max(0.010 * (d.rank + 1))          # rank 1 is the slow one
    ev_ms = d.max(9.0 * (d.rank + 1))
    assert abs(wall - 0.020) < 1e-12 and abs(ev_ms - 18.0) < 1e-12
    assert abs(d.sum(float(d.rank + 1)) - 3.0) < 1e-12
    agg = bench.aggregate(world=d.world, batch=256, steps=10, wall=wall, ev_ms=ev_ms, alg_bytes=5_532_357)
    assert agg["total_units"] == 2 * 256 * 10
    assert abs(agg["value"] - 2 * 256 * 10 / 0.020) < 1e-6
    assert abs(agg["launch_ms"] - 1.8) < 1e-12
    # frames shard across ranks with no overlap and nothing lost (strong-scaling helper)
    sizes = bench.shard_sizes(257, d.world)
    lo = sum(sizes[:d.rank])
    mine = set(range(lo, lo + sizes[d.rank]))
    total = d.sum(float(len(mine)))
    assert total == 257.0
    d.barrier()
    with open(os.path.join(out_dir, f"rank{d.rank}.ok"), "w") as fh:
        fh.write(f"{wall} {ev_ms}\n")
    d.close()


if __name__ == "__main__":
    main()
