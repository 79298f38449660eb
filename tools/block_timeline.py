"""Block timeline of k_plan_units (round 6): where a step's time goes BETWEEN the blocks -- idle tail, imbalance between the XCDs, block
durations by unit class.  Needs the experiment build (wrong-pixel-free: BEVW_EXPERIMENT=4 only adds two clock reads and one store per block):

    BEVW_BUILD_TAG=x4 BEVW_CFLAGS=-DBEVW_EXPERIMENT=4 python -m cameracalibration_amd.build
    BEVW_LIB_PATH=build_var/libbevwarp_x4.so python tools/block_timeline.py [--blend] [--batch 256] [--env BEVW_PLAN_NB=8 ...]

Every block records s_memrealtime (100 MHz) at its start and end, HW_ID and XCC_ID.  Prints, for the LAST of a few identical launches:
makespan, the sum of the block durations over slots x makespan (how full the chip was), per-XCD finish times, the tail (from the moment the
first of the 768 block slots goes idle for good to the end), and the block duration by unit class.
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--blend", action="store_true")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--dense", action="store_true")
    ap.add_argument("--env", nargs="*", default=[])
    ap.add_argument("--dump", default="")
    a = ap.parse_args()
    for kv in a.env:
        k, _, v = kv.partition("=")
        os.environ[k] = v
    from cameracalibration_amd import _ffi, workloads as W
    from cameracalibration_amd.SurroundBirdEyeView import surroundBEV as SB

    L = _ffi.lib()
    if not hasattr(L, "bevw_experiment_trace"):
        raise SystemExit("this libbevwarp has no bevw_experiment_trace: build with -DBEVW_EXPERIMENT=4 and point BEVW_LIB_PATH at it")
    cfg = W.CONFIG_S
    ns = SB.BevGenerator.get_args()
    for k, v in cfg.items():
        setattr(ns, k, v)
    bev = SB.BevGenerator(blend=a.blend, balance=False, rig=W.rig_s(), output_pitch="dense" if a.dense else "aligned")
    fw, fh, bh = cfg["FRAME_WIDTH"], cfg["FRAME_HEIGHT"], cfg["BEV_HEIGHT"]
    unique = W.synthetic_frames(2, fw, fh, seed=W.SEED)
    b_in = _ffi.DeviceBuffer(a.batch * unique[0].nbytes)
    b_out = _ffi.DeviceBuffer(a.batch * bh * bev.out_pitch * 3)
    for b in range(a.batch):
        b_in.upload(unique[b % 2], offset=b * unique[0].nbytes)
    for _ in range(4):
        bev.run_device(b_in.ptr, a.batch, None, b_out.ptr, out_bytes=a.batch * bev.out_image_bytes)
    bev.sync()
    rec = np.zeros((1 << 16, 4), np.uint64)   # t0, t1, hw_id | xcc_id << 32, class | pad << 32
    L.bevw_experiment_trace.argtypes = [C.c_void_p, C.c_size_t]
    assert L.bevw_experiment_trace(rec.ctypes.data, rec.nbytes) == 0
    t0, t1 = rec[:, 0].astype(np.int64), rec[:, 1].astype(np.int64)
    used = t1 > 0
    t0, t1 = t0[used], t1[used]
    hw = (rec[used, 2] & np.uint64(0xffffffff)).astype(np.int64)
    xcc = ((rec[used, 2] >> np.uint64(32)) & np.uint64(0xf)).astype(np.int64)
    cls_chunk = (rec[used, 3] & np.uint64(0xffffffff)).astype(np.int64)
    cls, chunk = cls_chunk & 0xff, cls_chunk >> 8
    tick_us = 0.01
    start = t0.min()
    t0u, t1u = (t0 - start) * tick_us, (t1 - start) * tick_us
    dur = t1u - t0u
    work = cls != 0xff
    span = t1u.max()
    print("blocks %d (%d with work), makespan %.1f us, sum of block durations %.1f us = %.1f slots busy on average" % (
        len(t0u), work.sum(), span, dur[work].sum(), dur[work].sum() / span))
    # blocks in flight over time
    ev = np.concatenate([np.stack([t0u[work], np.ones(work.sum())], 1), np.stack([t1u[work], -np.ones(work.sum())], 1)])
    ev = ev[np.argsort(ev[:, 0], kind="stable")]
    conc = np.cumsum(ev[:, 1])
    peak = conc.max()
    print("blocks in flight: peak %d" % peak)
    for frac in (0.05, 0.25, 0.5, 0.75, 0.9, 0.95, 0.98):
        i = np.searchsorted(ev[:, 0], frac * span)
        print("   at %4.0f %% of the makespan: %4d in flight" % (frac * 100, conc[min(i, len(conc) - 1)]))
    # the tail: from the last moment the chip ran at >= 90 % of its peak concurrency
    full = np.where(conc >= 0.9 * peak)[0]
    t_full_end = ev[full[-1], 0]
    idle_area = 0.0
    for k in range(full[-1], len(ev) - 1):
        idle_area += (peak - conc[k]) * (ev[k + 1, 0] - ev[k, 0])
    print("tail: concurrency drops below 90 %% of peak for good at %.1f us (%.1f %% of the makespan); idle slot-time in the tail = %.1f us x slots = %.1f %% of slots x makespan" % (
        t_full_end, 100 * t_full_end / span, idle_area, 100 * idle_area / (peak * span)))
    head = np.where(conc >= 0.9 * peak)[0][0]
    print("ramp: 90 %% of peak reached at %.1f us" % ev[head, 0])
    print("per XCD: blocks, first start, last end (us), busy slot-time")
    for x in sorted(set(xcc.tolist())):
        m = work & (xcc == x)
        if m.any():
            print("   xcc %d: %5d blocks  %7.1f .. %7.1f   %9.1f" % (x, m.sum(), t0u[m].min(), t1u[m].max(), dur[m].sum()))
    print("per chunk: first start, last end (us)")
    for c in sorted(set(chunk[work].tolist())):
        m = work & (chunk == c)
        print("   chunk %2d: %7.1f .. %7.1f  (%d blocks, xcc %s)" % (c, t0u[m].min(), t1u[m].max(), m.sum(), sorted(set(xcc[m].tolist()))))
    print("block duration by unit class (us): n, min, median, mean, p90, max, share of the busy slot-time")
    names = {0: "4x1", 1: "4x2", 2: "2x4", 3: "1x4", 4: "2x1d", 5: "1x4d", 6: "1x1d", 7: "4x4"}
    for c in sorted(set(cls[work].tolist())):
        m = work & (cls == c)
        dd = np.sort(dur[m])
        print("   class %d %-5s %6d  %6.1f %6.1f %6.1f %6.1f %6.1f   %5.1f %%" % (c, names.get(c, "?"), m.sum(), dd[0], dd[len(dd) // 2], dd.mean(), dd[int(0.9 * len(dd))], dd[-1],
                                                                                   100 * dd.sum() / dur[work].sum()))
    if a.dump:
        np.save(a.dump, np.stack([t0u, t1u, xcc, hw, cls, chunk], 1))


if __name__ == "__main__":
    main()
