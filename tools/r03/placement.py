"""Placement study of config 3: one process, the engine built once; the frame / output buffers are
(A) freed and re-allocated between trials (what separate bench processes draw), (B) carved out of ONE allocation at different
sub-offsets (same physical pages: only the relative alignment of the two streams changes).
    python tools/r03/placement.py [--trials N] [--mode realloc|offset] [--steps K]
Prints per trial: pointers, median / min of K per-step HIP-event times."""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cameracalibration_amd import _ffi, workloads as W  # noqa: E402
from cameracalibration_amd.SurroundBirdEyeView import surroundBEV as SB  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--trials", type=int, default=10)
ap.add_argument("--steps", type=int, default=16)
ap.add_argument("--mode", default="realloc")
ap.add_argument("--batch", type=int, default=256)
a = ap.parse_args()

cfg, rig = W.CONFIG_S, W.rig_s()
ns = SB.BevGenerator.get_args()
for k, v in cfg.items():
    setattr(ns, k, v)
bev = SB.BevGenerator(rig=rig)
fw, fh, bw, bh = cfg["FRAME_WIDTH"], cfg["FRAME_HEIGHT"], cfg["BEV_WIDTH"], cfg["BEV_HEIGHT"]
batch = a.batch
unique = W.synthetic_frames(2, fw, fh, seed=W.SEED)
set_bytes, img_bytes = unique[0].nbytes, bh * bw * 3


def fill(ptr_buf, off):
    for b in range(batch):
        ptr_buf.upload(unique[b % 2], off + b * set_bytes)


def measure(p_in, p_out):
    for _ in range(3):
        bev.run_device(p_in, batch, None, p_out)
    bev.sync()
    for i in range(a.steps + 1):
        bev.timer_mark(i)
        if i < a.steps:
            bev.run_device(p_in, batch, None, p_out)
    laps = sorted(bev.timer_between(i, i + 1) for i in range(a.steps))
    return laps[len(laps) // 2], laps[0]


if a.mode == "realloc":
    dummies = []
    for t in range(a.trials):
        d_in = _ffi.DeviceBuffer(batch * set_bytes, 0)
        d_out = _ffi.DeviceBuffer(batch * img_bytes, 0)
        fill(d_in, 0)
        med, mn = measure(d_in.ptr, d_out.ptr)
        print("realloc trial %2d in %#x out %#x median %.4f min %.4f ms" % (t, d_in.ptr, d_out.ptr, med, mn), flush=True)
        d_in.free(); d_out.free()
        dummies.append(_ffi.DeviceBuffer((t + 1) * 37 * 1024 * 1024 + 4096, 0))
elif a.mode == "arena":
    # ONE allocation of 24 GiB; the two buffers at offsets that are GiB apart: does the position inside the physical heap matter?
    G = 1 << 30
    arena = _ffi.DeviceBuffer(48 * G, 0)
    pairs = [(i, i + dd) for i in (0, 8, 16, 24, 3) for dd in (4, 8, 12, 16, 20, 24, 32) if i + dd + 1 <= 48] + [(20, 0), (36, 4), (40, 8), (44, 28)]
    for gi, go in pairs[:a.trials]:
        fill(arena, gi * G)
        med, mn = measure(arena.ptr + gi * G, arena.ptr + go * G)
        print("arena in+%2d GiB out+%2d GiB (base %#x) median %.4f min %.4f ms" % (gi, go, arena.ptr, med, mn), flush=True)
elif a.mode == "delta":
    # ONE allocation; input fixed, output at 8 GiB + delta: do the address bits between 2 MiB and 1 GiB of the two streams interact?
    G, M = 1 << 30, 1 << 20
    arena = _ffi.DeviceBuffer(16 * G, 0)
    fill(arena, 0)
    for dl in [0, 2, 4, 8, 16, 32, 64, 96, 128, 192, 256, 320, 384, 448, 512, 576, 640, 704, 768, 832, 896, 960, 1024, 1536][:a.trials]:
        med, mn = measure(arena.ptr, arena.ptr + 8 * G + dl * M)
        print("delta out = in + 8 GiB + %4d MiB (base %#x) median %.4f min %.4f ms" % (dl, arena.ptr, med, mn), flush=True)
    for dl in [0, 7, 64, 100, 333, 512, 777][:a.trials]:     # the input moved instead (re-filled)
        fill(arena, dl * M)
        med, mn = measure(arena.ptr + dl * M, arena.ptr + 8 * G)
        print("delta in + %4d MiB, out = + 8 GiB median %.4f min %.4f ms" % (dl, med, mn), flush=True)
elif a.mode == "flags":
    # hipExtMallocWithFlags: default / physically contiguous (hipDeviceMallocContiguous) -- is the spread a matter of page contiguity?
    hip = C.CDLL("libamdhip64.so")
    hip.hipExtMallocWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
    hip.hipFree.argtypes = [C.c_void_p]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    dummies = []
    for t in range(a.trials):
        for flag, name in ((0x4, "contiguous"), (0x0, "default")):
            p_in, p_out = C.c_void_p(), C.c_void_p()
            e1 = hip.hipExtMallocWithFlags(C.byref(p_in), batch * set_bytes, flag)
            e2 = hip.hipExtMallocWithFlags(C.byref(p_out), batch * img_bytes, flag)
            if e1 or e2:
                print("flags %s: hipExtMallocWithFlags failed (%d, %d)" % (name, e1, e2), flush=True)
                continue
            for b in range(batch):
                u = unique[b % 2]
                hip.hipMemcpy(p_in.value + b * set_bytes, u.ctypes.data, u.nbytes, 1)
            med, mn = measure(p_in.value, p_out.value)
            print("flags trial %2d %-10s in %#x out %#x median %.4f min %.4f ms" % (t, name, p_in.value, p_out.value, med, mn), flush=True)
            hip.hipFree(p_in); hip.hipFree(p_out)
        dummies.append(_ffi.DeviceBuffer((t + 1) * 37 * 1024 * 1024 + 4096, 0))
elif a.mode == "cycle":
    # three buffer pairs per trial: one pair re-processed every step (what bench.py does) against the three pairs in turn (4.8 GB x 3 between
    # two visits of a line): does the Infinity Cache carry lines from one step to the next in the fast placements?
    dummies = []
    for t in range(a.trials):
        pairs = []
        for k in range(3):
            d_in = _ffi.DeviceBuffer(batch * set_bytes, 0)
            d_out = _ffi.DeviceBuffer(batch * img_bytes, 0)
            fill(d_in, 0)
            pairs.append((d_in, d_out))
        singles = [measure(p[0].ptr, p[1].ptr)[0] for p in pairs]
        for _ in range(3):
            for p in pairs:
                bev.run_device(p[0].ptr, batch, None, p[1].ptr)
        bev.sync()
        n = 3 * (a.steps // 3 + 1)
        for i in range(n + 1):
            bev.timer_mark(i)
            if i < n:
                p = pairs[i % 3]
                bev.run_device(p[0].ptr, batch, None, p[1].ptr)
        laps = [bev.timer_between(i, i + 1) for i in range(n)]
        per = [sorted(laps[k::3])[len(laps[k::3]) // 2] for k in range(3)]
        print("cycle trial %2d: each pair alone %s | in turn %s" % (t, " ".join("%.4f" % v for v in singles), " ".join("%.4f" % v for v in per)), flush=True)
        # which buffer carries the property?  input of pair i with output of pair j
        cross = [[measure(pairs[i][0].ptr, pairs[j][1].ptr)[0] for j in range(3)] for i in range(3)]
        print("      cross (rows: input of pair i, columns: output of pair j): " + " | ".join(" ".join("%.4f" % v for v in row) for row in cross), flush=True)
        for p in pairs:
            p[0].free(); p[1].free()
        dummies.append(_ffi.DeviceBuffer((t + 1) * 37 * 1024 * 1024 + 4096, 0))
elif a.mode == "vmm":
    # virtually contiguous buffers backed by physical chunks that are mapped in SHUFFLED order (hipMemCreate / hipMemMap): if physical
    # contiguity is what makes a placement slow, a permutation at chunk granularity should make every draw fast
    import random
    hip = C.CDLL("libamdhip64.so")

    class Loc(C.Structure):
        _fields_ = [("type", C.c_int), ("id", C.c_int)]

    class AFlags(C.Structure):
        _fields_ = [("compressionType", C.c_ubyte), ("gpuDirectRDMACapable", C.c_ubyte), ("usage", C.c_ushort)]

    class Prop(C.Structure):
        _fields_ = [("type", C.c_int), ("requestedHandleType", C.c_int), ("location", Loc), ("win32HandleMetaData", C.c_void_p), ("allocFlags", AFlags)]

    class Access(C.Structure):
        _fields_ = [("location", Loc), ("flags", C.c_int)]

    prop = Prop()
    prop.type, prop.requestedHandleType, prop.location.type, prop.location.id = 1, 0, 1, 0
    gran = C.c_size_t()
    hip.hipMemGetAllocationGranularity.argtypes = [C.POINTER(C.c_size_t), C.POINTER(Prop), C.c_int]
    print("granularity rc", hip.hipMemGetAllocationGranularity(C.byref(gran), C.byref(prop), 0), gran.value, flush=True)
    hip.hipMemAddressReserve.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_size_t, C.c_void_p, C.c_ulonglong]
    hip.hipMemCreate.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.POINTER(Prop), C.c_ulonglong]
    hip.hipMemMap.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_ulonglong]
    hip.hipMemSetAccess.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(Access), C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]

    def vmm_alloc(nbytes, chunk, shuffle, seed):
        chunk = max(chunk, gran.value)
        n = (nbytes + chunk - 1) // chunk
        base = C.c_void_p()
        assert hip.hipMemAddressReserve(C.byref(base), n * chunk, 0, None, 0) == 0
        handles = []
        for _ in range(n):
            h = C.c_void_p()
            rc = hip.hipMemCreate(C.byref(h), chunk, C.byref(prop), 0)
            assert rc == 0, rc
            handles.append(h)
        order = list(range(n))
        if shuffle:
            random.Random(seed).shuffle(order)
        for i, k in enumerate(order):
            assert hip.hipMemMap(base.value + i * chunk, chunk, 0, handles[k], 0) == 0
        acc = Access()
        acc.location.type, acc.location.id, acc.flags = 1, 0, 3
        assert hip.hipMemSetAccess(base, n * chunk, C.byref(acc), 1) == 0
        return base.value

    import time
    d_in = _ffi.DeviceBuffer(batch * set_bytes, 0)
    fill(d_in, 0)
    d_out = _ffi.DeviceBuffer(batch * img_bytes, 0)
    med, mn = measure(d_in.ptr, d_out.ptr)
    print("vmm reference: both buffers from hipMalloc median %.4f min %.4f ms" % (med, mn), flush=True)
    for t in range(a.trials):
        for chunk_kb, shuffle in ((2048, True), (2048, False), (4096, True), (8192, True), (2048, True), (32768, True), (2048, True)):
            t0 = time.time()
            p_out = vmm_alloc(batch * img_bytes, chunk_kb << 10, shuffle, 200 + t)
            t1 = time.time() - t0
            med, mn = measure(d_in.ptr, p_out)
            print("vmm trial %d OUTPUT in chunks of %4d KiB %-9s (mapped in %.1f s) median %.4f min %.4f ms" % (t, chunk_kb, "shuffled" if shuffle else "in order", t1, med, mn), flush=True)
            # (buffers are left mapped: the study process ends soon)
elif a.mode == "joint":
    # both buffers inside ONE allocation per trial, re-allocated between trials
    dummies = []
    for t in range(a.trials):
        n_in = (batch * set_bytes + (2 << 20) - 1) & ~((2 << 20) - 1)
        buf = _ffi.DeviceBuffer(n_in + batch * img_bytes, 0)
        fill(buf, 0)
        med, mn = measure(buf.ptr, buf.ptr + n_in)
        print("joint trial %2d base %#x median %.4f min %.4f ms" % (t, buf.ptr, med, mn), flush=True)
        buf.free()
        dummies.append(_ffi.DeviceBuffer((t + 1) * 37 * 1024 * 1024 + 4096, 0))
else:
    slack = 64 << 20
    d_in = _ffi.DeviceBuffer(batch * set_bytes + slack, 0)
    d_out = _ffi.DeviceBuffer(batch * img_bytes + slack, 0)
    offs = [0, 256, 4096, 8192, 65536, 1 << 20, (1 << 20) + 4096, 2 << 20, 3 << 20, 32 << 20]
    last = None
    for oi in offs[:a.trials]:
        if last != oi:
            fill(d_in, oi)
            last = oi
        for oo in (0, 4096, 1 << 20):
            med, mn = measure(d_in.ptr + oi, d_out.ptr + oo)
            print("offset in+%#x out+%#x median %.4f min %.4f ms" % (oi, oo, med, mn), flush=True)
