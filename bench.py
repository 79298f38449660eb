#!/usr/bin/env python3
"""bench.py -- stitched-BEV throughput of the HIP engine on N MI355X GPUs (one process per GPU).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME] [--batch B] [--schedule auto|pixel|plan]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path (BevGenerator.__call__ for a whole batch, surroundBEV.py:312-325) over one
batch of synthetic frames that is ALREADY RESIDENT IN HBM (bevw_run_device).  Frames shard across ranks with no
data-path collective (each rank owns its own batch and a full copy of the static tables): weak scaling.  Rank 0
prints ONE JSON line.  Workloads = BASELINE.json configs:

    direct_stitch_b256   4-cam direct stitch 1280x960 -> 1080x1080, batch 256 per GPU   (default: the metric's config)
    blend_balance_b256   same sizes, BevGenerator(blend=True, balance=True)
    undistort_b64        single fisheye undistort remap 1280x960, batch 64
    blend_4k             4-cam 3840x2160 -> 1080x1080 blend, batch 32
    jpeg_decode_b64 / jpeg_encode_b64 / jpeg_bev_jpeg_b64   the JPEG wire format either side of the path on the GPU (row f4)

`roofline.achieved` = algorithmic bytes of the workload (cameracalibration_amd/workloads.py) x units
per launch / the launch's average duration measured with HIP events on the engine's own stream.
`cpu_baseline` = the CPU oracle (oracle/, reference operation order, OpenMP over all host cores) timed on a bounded
sample of the same workload on rank 0 at N=1 -- a reported baseline, not the thing measured.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)

WORKLOADS = {
    "direct_stitch_b256": dict(kind="bev", cfg="S", blend=False, balance=False, batch=256, unit="frames/s",
                               metric="stitched BEV frames/sec (4-cam 1280x960->1080x1080)"),
    "blend_b256": dict(kind="bev", cfg="S", blend=True, balance=False, batch=256, unit="frames/s",
                       metric="stitched BEV frames/sec (4-cam 1280x960->1080x1080, blend)"),
    "blend_balance_b256": dict(kind="bev", cfg="S", blend=True, balance=True, batch=256, unit="frames/s",
                               metric="stitched BEV frames/sec (4-cam 1280x960->1080x1080, blend+balance)"),
    "undistort_b64": dict(kind="undistort", batch=64, unit="images/s",
                          metric="fisheye undistort remap images/sec (1280x960 u8)"),
    "blend_4k": dict(kind="bev", cfg="4K", blend=True, balance=False, batch=32, unit="frames/s",
                     metric="stitched BEV frames/sec (4-cam 3840x2160->1080x1080, blend)"),
    # The table-free projection mode.  The first two evaluate the camera model ONCE PER HANDLE on the GPU (k_analytic_map, fp32 / fp64)
    # into a wide unit plan with 21-bit fractions and then run the unit schedule with fp32 interpolation (k_plan_unit_wide): per frame they cost what a
    # table with finer fractions costs, NOT a projection.  The third is north_star's wording taken literally: k_stitch_perpixel, one thread per
    # pixel and frame -- inverse homography + K / D fisheye model + fp32 bilinear sample evaluated per output pixel AND per frame, no table anywhere.
    "direct_stitch_analytic_f32_b64": dict(kind="bev", cfg="S", blend=False, balance=False, batch=64, unit="frames/s", projection="analytic_f32",
                                           metric="stitched BEV frames/sec (4-cam 1280x960->1080x1080, analytic fp32 projection evaluated once per handle, 21-bit fractions, fp32 interpolation)"),
    "direct_stitch_analytic_f64_b64": dict(kind="bev", cfg="S", blend=False, balance=False, batch=64, unit="frames/s", projection="analytic",
                                           metric="stitched BEV frames/sec (4-cam 1280x960->1080x1080, analytic fp64 projection evaluated once per handle, 21-bit fractions, fp32 interpolation)"),
    "direct_stitch_analytic_perpixel_b64": dict(kind="bev", cfg="S", blend=False, balance=False, batch=64, unit="frames/s", projection="analytic_f32",
                                                env={"BEVW_ANALYTIC_UNITS": "0", "BEVW_ANALYTIC_FRAMES": "1"},
                                                metric="stitched BEV frames/sec (4-cam 1280x960->1080x1080, fused per-output-pixel kernel: inverse homography + fisheye "
                                                       "projection in fp32 evaluated per pixel per frame, no table)"),
    # BASELINE config 5 in its camera-per-GPU form: ranks own cameras, parts travel over RCCL
    # send/recv, the stitch rank rotates.  1, 2 or a multiple of 4 ranks; every group of 4 ranks is a replica.
    # the JPEG wire format either side of the path, on the GPU (cameracalibration_amd/imgcodecs.py).  Inputs resident =
    # the staged (parsed + un-stuffed) compressed streams / the device images; the timed region is the kernels only.
    "jpeg_decode_b64": dict(kind="jpeg", mode="decode", cfg="S", batch=64, unit="images/s",
                            metric="camera JPEG files decoded into frame sets per second (1280x960 baseline 4:2:0 -> BGR)"),
    "jpeg_encode_b64": dict(kind="jpeg", mode="encode", cfg="S", batch=64, unit="images/s",
                            metric="BEV images encoded to .jpg files per second (1080x1080 BGR -> baseline 4:2:0, quality 95)"),
    "jpeg_bev_jpeg_b64": dict(kind="jpeg", mode="pipeline", cfg="S", batch=64, unit="frames/s", blend=False, balance=False,
                              metric="stitched BEV frames/sec, compressed in and out (4 x 1280x960 .jpg -> 1080x1080 .jpg)"),
    "blend_4k_camera_shard": dict(kind="camera", cfg="4K", blend=True, balance=False, batch=32, unit="frames/s",
                                  metric="stitched BEV frames/sec (4-cam 3840x2160->1080x1080, blend, camera per GPU)"),
}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="direct_stitch_b256", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="frames (4-camera sets) per GPU per step; 0 = workload default")
    ap.add_argument("--schedule", default="auto", choices=["auto", "pixel", "plan"])
    ap.add_argument("--unique-sets", type=int, default=2, help="distinct synthetic frame sets replicated over the batch")
    ap.add_argument("--placements", type=int, default=0,
                    help="buffer placements: the frame / output buffers are freed and re-allocated this many times and K steps are timed "
                         "on each; the MEDIAN placement is reported (0 = 5 for the single-engine workloads, 1 for the camera-shard one)")
    ap.add_argument("--output-pitch", default="aligned", type=lambda v: v if v in ("aligned", "dense") else int(v),
                    help="row pitch of the device-resident BEV images (bevw_set_output_pitch): aligned = rows of whole 64-byte sectors "
                         "(1080 -> 1088 pixels, cv::cuda::GpuMat style; what BevGenerator's default output_pitch='auto' selects on the tile "
                         "plan), dense = the reference's host layout; the other layout is measured too and reported beside the headline")
    ap.add_argument("--single-layout", action="store_true", help="skip the measurement of the other device-image layout (profiling runs)")
    ap.add_argument("--jpeg-source", default="synthetic", choices=["synthetic", "repo"],
                    help="jpeg_decode_b64 only: 'repo' decodes the reference's own four camera files (tests/golden/repo_rig.npz, 1280x1024, real scenes with "
                         "flat areas -- long mis-phased stretches for the parallel Huffman decoder) replicated over the batch, instead of synthetic files")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-live-traffic", action="store_true", help="skip the two rocprofv3 --pmc child runs behind roofline.traffic (falls back to profiles/hbm_traffic.json)")
    ap.add_argument("--no-f4", action="store_true", help="skip the JPEG summary (`f4`) the default single-GPU run appends to its line")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args()


class Dist:
    """Barrier / max over ranks.  torch.distributed (backend nccl == RCCL) only when WORLD_SIZE > 1."""

    def __init__(self):
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.t = None
        if self.world > 1:
            import torch
            import torch.distributed as dist

            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            # BEVW_BENCH_BACKEND=gloo: every rank on one GPU with host-staged exchange (validation on a 1-GPU box)
            use_gpu = torch.cuda.is_available() and os.environ.get("BEVW_BENCH_BACKEND", "nccl") == "nccl"
            if use_gpu:
                torch.cuda.set_device(self.local_rank)
            dist.init_process_group("nccl" if use_gpu else "gloo")
            self.torch, self.dist = torch, dist
            self.dev = torch.device("cuda", self.local_rank) if use_gpu else torch.device("cpu")

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()

    def max(self, v: float) -> float:
        if self.world == 1:
            return v
        t = self.torch.tensor([v], dtype=self.torch.float64, device=self.dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum(self, v: float) -> float:
        if self.world == 1:
            return v
        t = self.torch.tensor([v], dtype=self.torch.float64, device=self.dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())

    def close(self):
        if self.world > 1:
            self.dist.destroy_process_group()


def shard_sizes(total_units: int, world: int):
    """Units (frames) per rank when a fixed total is split (strong scaling helper; also used by the gloo test)."""
    base, rem = divmod(total_units, world)
    return [base + (1 if r < rem else 0) for r in range(world)]


def aggregate(world: int, batch: int, steps: int, wall: float, ev_ms: float, alg_bytes: int) -> dict:
    """Whole-job numbers from the max-over-ranks wall time (s) and HIP-event time (ms) of `steps` steps."""
    total_units = batch * steps * world
    launch_ms = ev_ms / steps
    return {"total_units": total_units, "value": total_units / wall, "launch_ms": launch_ms,
            "achieved_gbs": alg_bytes * batch / (launch_ms * 1e-3) / 1e9}


def static_traffic(workload: str, batch: int):
    """(HBM bytes per launch, fetch, write, source) from the committed rocprofv3 PMC passes (profiles/hbm_traffic.json: separate --pmc
    FETCH_SIZE / WRITE_SIZE runs of this same command, corrected with profiles/pmc_calibration.json), or Nones.  A STATIC figure of the
    collection it came from -- the fallback when the live measurement below cannot run."""
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json"))).get(workload)
        if rec and rec.get("units_per_launch") == batch:
            return rec["fetch_bytes"] + rec["write_bytes"], rec["fetch_bytes"], rec["write_bytes"], (
                "profiles/hbm_traffic.json (static, rocprofv3 --pmc passes of round %s%s)" % (rec.get("round", 1), ", counters calibrated on known byte counts" if rec.get("corrected") else ""))
    except (OSError, ValueError):
        pass
    return None, None, None, None


def live_traffic(a, batch: int):
    """HBM bytes per launch of THIS build on THIS box: two child runs of this script (3 steps + 1 warm-up, one placement, the same layout)
    under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes: the two counters do not fit the TCC's slots together;
    no tracing beside the counters), summed over the step's kernels and corrected as tools/summarize_pmc.py does (the guide's gfx950
    note: FETCH_SIZE tallies 128-byte requests at 64 bytes; factors measured on known byte counts, profiles/pmc_calibration.json).
    Returns (total, fetch, write, source) or None when rocprofv3 is missing / fails (the caller falls back to the static file)."""
    import glob
    import importlib.util
    import shutil
    import subprocess
    import tempfile
    profiled = "rocprofiler" in os.environ.get("LD_PRELOAD", "") or bool(os.environ.get("ROCP_TOOL_LIBRARIES"))   # already under rocprofv3
    if os.environ.get("BEVW_BENCH_CHILD") or a.no_live_traffic or profiled or shutil.which("rocprofv3") is None:
        return None
    spec = importlib.util.spec_from_file_location("bevw_summarize_pmc", os.path.join(ROOT, "tools", "summarize_pmc.py"))
    S = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(S)
    steps, warmup, files = 3, 1, {}
    env = dict(os.environ, BEVW_BENCH_CHILD="1", TMPDIR="/tmp")
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="bevw_pmc_", dir="/tmp")
        cmd = ["rocprofv3", "--pmc", counter, "--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__), "--workload", a.workload,
               "--steps", str(steps), "--warmup", str(warmup), "--placements", "1", "--single-layout", "--no-cpu-baseline", "--no-f4",
               "--output-pitch", str(a.output_pitch), "--schedule", a.schedule, "--unique-sets", str(a.unique_sets)] + (["--batch", str(a.batch)] if a.batch else [])
        try:
            r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=240)
            found = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not found:
                return None
            files[counter] = found[0]
        except (OSError, subprocess.SubprocessError):
            return None
    try:
        tf, tw, rows = S.per_launch_traffic(files["FETCH_SIZE"], files["WRITE_SIZE"], steps + warmup)
    except Exception:
        return None
    finally:
        for f in files.values():
            shutil.rmtree(os.path.dirname(os.path.dirname(f)), ignore_errors=True)
    if not rows or tf <= 0 or tw <= 0:
        return None
    fb, wb = int(tf * 1024), int(tw * 1024)
    return fb + wb, fb, wb, ("live: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE child runs of this command on this box (%d launches each), "
                             "corrected with profiles/pmc_calibration.json%s" % (steps + warmup, "" if S.CAL else " (MISSING: uncorrected)"))


def upload_replicated(buf, unique: np.ndarray, batch: int):
    per = unique[0].nbytes
    for b in range(batch):
        buf.upload(unique[b % unique.shape[0]], offset=b * per)


class TimedOracle:
    """The oracle source compiled -O3 -march=native ON THIS HOST for the timed cpu_baseline leg only;
    parity everywhere else uses the -O2 build.  Same set_threads / usable_cores surface as the oracle module."""

    def __init__(self, O):
        path = O.build_timed()
        self.lib = O.load(path)
        self.flags = "-O3 -march=native -fopenmp -ffp-contract=off" if "_timed" in path else "-O2 -fopenmp -ffp-contract=off"
        self.usable_cores = O.usable_cores

    def set_threads(self, n):
        self.lib.orc_set_threads(int(n))


def timed_loop(fn, seconds, cap):
    n, t0 = 0, time.perf_counter()
    while True:
        fn(n)
        n += 1
        dt = time.perf_counter() - t0
        if dt >= seconds or n >= cap:
            return n, dt


def timed_rate(T, threads, fn, seconds, cap) -> float:
    T.set_threads(threads)
    fn(0)
    n, dt = timed_loop(fn, seconds, cap)
    return n / dt


def pick_threads(O, fn) -> int:
    """OpenMP thread count for the CPU baseline: the fastest of a few candidates up to the cores this process may use
    (a container can show far more cores than its quota allows; oversubscription thrashes)."""
    lim = O.usable_cores(256)
    cands = sorted({c for c in (1, 4, 8, 16, 32, 64, 128, lim) if c <= lim})
    best, best_t = 1, float("inf")
    for c in cands:
        O.set_threads(c)
        fn(0)
        t0 = time.perf_counter()
        fn(1); fn(2)
        dt = time.perf_counter() - t0
        if dt < best_t:
            best, best_t = c, dt
        if dt > 4 * best_t:
            break
    O.set_threads(best)
    return best


def cpu_baseline_bev(w, cfg, rig, unique, seconds):
    """Oracle in the reference's operation order (oracle.RefBevGenerator.make_fast_call), all host cores."""
    from oracle import oracle as O

    O.build()
    ref = O.RefBevGenerator(rig, cfg, blend=w["blend"], balance=w["balance"])
    frames = [np.ascontiguousarray(unique[0][c]) for c in range(4)]
    want = ref.make_fast_call()(frames).copy()            # parity build (-O2), the checker
    T = TimedOracle(O)                                     # the same source, -O3 -march=native, built on this host
    call = ref.make_fast_call(T.lib)
    assert np.array_equal(call(frames), want), "the -O3 -march=native oracle build changed the arithmetic"
    one = timed_rate(T, 1, lambda i: call(frames), min(seconds / 3, 4.0), 400)
    cores = pick_threads(T, lambda i: call(frames))
    sets = [[np.ascontiguousarray(unique[k][c]) for c in range(4)] for k in range(unique.shape[0])]
    n, dt = timed_loop(lambda i: call(sets[i % len(sets)]), seconds, 2000)
    return {"value": n / dt, "unit": w["unit"], "cores": cores, "kind": "port", "value_1_thread": one,
            "build": T.flags,
            "sample": f"{n} stitched frames in {dt:.1f} s (oracle/bevoracle.c orc_bev_call, OpenMP {cores} threads of "
                      f"{os.cpu_count()} visible, reference op order: remap x4 -> mask -> 3 sat-adds); 1 thread: {one:.1f} {w['unit']}"}


def cpu_baseline_undistort(w, K, D, ucfg, unique, seconds):
    from oracle import oracle as O

    O.build()
    fw, fh = ucfg["FRAME_WIDTH"], ucfg["FRAME_HEIGHT"]
    Kd = O.camera_mat_dst(K, fw, fh, ucfg["FOCAL_SCALE"], ucfg["SIZE_SCALE"])
    m1, m2 = O.fisheye_init_undistort_rectify_map(K, D, Kd, (int(fw * ucfg["SIZE_SCALE"]), int(fh * ucfg["SIZE_SCALE"])))
    img = np.ascontiguousarray(unique[0][0])
    want = O.remap(img, m1, m2)
    T = TimedOracle(O)
    out = np.empty_like(want)

    def remap(_i=0):
        T.lib.orc_remap_u8(img.ctypes.data, img.shape[1], img.shape[0], 3, m1.ctypes.data, m2.ctypes.data, m2.shape[1], m2.shape[0],
                           out.ctypes.data)
    remap()
    assert np.array_equal(out, want), "the -O3 -march=native oracle build changed the arithmetic"
    one = timed_rate(T, 1, remap, min(seconds / 3, 4.0), 2000)
    cores = pick_threads(T, remap)
    n, dt = timed_loop(remap, seconds, 5000)
    return {"value": n / dt, "unit": w["unit"], "cores": cores, "kind": "port", "value_1_thread": one, "build": T.flags,
            "sample": f"{n} images in {dt:.1f} s (oracle orc_remap_u8, OpenMP {cores} threads); 1 thread: {one:.1f} {w['unit']}"}


def camera_like_jpegs(n_files: int, width: int, height: int, seed: int, quality: int = 90):
    """Synthetic camera FILES: the smooth synthetic frames, 2 x 2 box-filtered (sensor-like noise level), written by libjpeg-turbo (Pillow)
    at `quality`, 4:2:0 -- input generation only, outside every timed region."""
    import io
    from PIL import Image
    from cameracalibration_amd import workloads as W

    frames = W.synthetic_frames((n_files + 3) // 4, width, height, seed=seed).reshape(-1, height, width, 3)[:n_files]
    files = []
    for f in frames:
        g = f.astype(np.int32)
        g = ((g + np.roll(g, 1, 0) + np.roll(g, 1, 1) + np.roll(np.roll(g, 1, 0), 1, 1) + 2) // 4).astype(np.uint8)
        b = io.BytesIO()
        Image.fromarray(np.ascontiguousarray(g[:, :, ::-1])).save(b, "JPEG", quality=quality, subsampling=2)
        files.append(b.getvalue())
    return files


def jpeg_cpu_baseline(mode, files, images, seconds):
    """libjpeg-turbo itself (Pillow's build, SIMD) on the host cores -- the library cv2.imread / cv2.imwrite wrap, so kind = reference.
    Pillow releases the GIL inside the codec: one thread per usable core."""
    import io
    from concurrent.futures import ThreadPoolExecutor
    from PIL import Image
    from oracle import oracle as O

    cores = O.usable_cores(64)
    pil_images = [Image.fromarray(np.ascontiguousarray(im[:, :, ::-1])) for im in images] if images is not None else None

    def dec(i):
        return np.asarray(Image.open(io.BytesIO(files[i % len(files)])).convert("RGB")).shape

    def enc(i):
        b = io.BytesIO()
        pil_images[i % len(pil_images)].save(b, "JPEG", quality=95, subsampling=2)
        return b.tell()
    fn = dec if mode == "decode" else enc
    t0 = time.perf_counter()
    fn(0)
    one = 1.0 / (time.perf_counter() - t0)
    n, t0 = 0, time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        while time.perf_counter() - t0 < seconds:
            list(ex.map(fn, range(n, n + 4 * cores)))
            n += 4 * cores
    dt = time.perf_counter() - t0
    many = n / dt
    if many < 1.2 * one:   # Pillow's encoder holds the GIL most of the time: threads do not scale, so the honest figure is one core's
        n1, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < seconds / 2:
            fn(n1)
            n1 += 1
        dt1 = time.perf_counter() - t0
        return {"value": n1 / dt1, "unit": "images/s", "cores": 1, "kind": "reference", "value_threads": many,
                "sample": f"{n1} images {mode}d in {dt1:.1f} s by Pillow's libjpeg-turbo (SIMD) on ONE thread; {cores} Python threads reach "
                          f"{many:.0f} images/s (the GIL is held around the codec call), so this is a per-core figure"}
    return {"value": many, "unit": "images/s", "cores": cores, "kind": "reference", "value_1_thread": one,
            "sample": f"{n} images {mode}d in {dt:.1f} s by Pillow's libjpeg-turbo (SIMD), {cores} threads"}


VALU_CLOCK_HZ = 2.4e9        # MI355X peak engine clock
# wave64 VALU instructions per second the chip can issue: 1024 SIMDs of 32 lanes; 2 clocks for a full-rate instruction (add / and / or / mov /
# fp32 mul, add, fma), 4 for the rest (shifts, 24-bit mads, perm / bfe, dot products, conversions, compares) -- MI355X_MICROARCH.md "Wave
# scheduling", tools/valu_rates.hip.  A kernel's own peak follows from its instruction mix (tools/valu_mix.py -> profiles/jpeg_valu.json:
# peak_ginst); without a mix the all-full-rate figure is the (unreachable) upper bound.
VALU_PEAK_GINST_FULL_RATE = 256 * 4 * VALU_CLOCK_HZ / 2 / 1e9


def jpeg_valu_profile(workload: str):
    """Static per-unit instruction counts of the JPEG workloads from the committed rocprofv3 PMC pass (profiles/jpeg_valu.json: SQ_INSTS_VALU
    summed over the kernels of one step / the units of the step), or None.  Like roofline.traffic it is a figure of the round it was
    collected in, not a measurement of this run."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "jpeg_valu.json"))).get(workload)
    except (OSError, ValueError):
        return None


def jpeg_measure(a, d, w, dev, workload, steps, warmup, cpu_seconds, host_api=True):
    """Row f4 workloads.  One step = one batch through the JPEG kernels with the inputs resident in HBM: decode = the files' entropy-coded bytes
    AS THEY ARE IN THE FILES -> frame sets (un-stuffing, entropy decoding, inverse DCT, colour: everything a decode needs is inside the timed
    region); encode = device images -> complete files in HBM; pipeline = decode + stitch + encode chained by events on their streams."""
    from cameracalibration_amd import _ffi, imgcodecs, workloads as W

    _ffi.prefer_hw_queues(8)   # the decoder's slice streams and the engine's streams on hardware queues of their own (before HIP initialises)
    from cameracalibration_amd.SurroundBirdEyeView import surroundBEV as SB

    cfg = W.CONFIG_S
    batch = a.batch or w["batch"]
    fw, fh, bw, bh = cfg["FRAME_WIDTH"], cfg["FRAME_HEIGHT"], cfg["BEV_WIDTH"], cfg["BEV_HEIGHT"]
    mode = w["mode"]
    codec = imgcodecs.JpegCodec(dev)
    uniq_files = camera_like_jpegs(8, fw, fh, seed=W.SEED + d.rank)
    if a.jpeg_source == "repo":
        if mode != "decode":
            raise SystemExit("--jpeg-source repo is for jpeg_decode_b64")
        z = np.load(os.path.join(ROOT, "tests", "golden", "repo_rig.npz"))
        uniq_files = [z[f"{n}_img"].tobytes() for n in W.CAMERA_NAMES]
        probe = imgcodecs.probe(uniq_files[0])
        fw, fh = probe["width"], probe["height"]
    files = [uniq_files[i % len(uniq_files)] for i in range(batch * 4)]
    uniq_bev = W.synthetic_frames(1, bw, bh, seed=W.SEED + 17 + d.rank)[0]          # four BEV-sized camera-like images
    bev_images = np.stack([uniq_bev[i % 4] for i in range(batch)])
    d_frames = _ffi.DeviceBuffer(batch * 4 * fh * fw * 3, dev)
    d_bev = _ffi.DeviceBuffer(batch * bh * bw * 3, dev)
    extra = {"frame": [fw, fh], "bev": [bw, bh], "jpeg_in": ("baseline 4:2:0, %d bytes per file (%s)" % (
        sum(len(f) for f in uniq_files) // len(uniq_files), "the reference's own camera files, %dx%d" % (fw, fh) if a.jpeg_source == "repo"
        else "synthetic camera-like frames, quality 90")), "jpeg_out": "baseline 4:2:0 quality 95 (cv2.imwrite's defaults)"}
    stage_ms = None
    bev = None
    if mode in ("decode", "pipeline"):
        codec.decode_stage(files)
        codec.sync()
        t0 = time.perf_counter()
        for _ in range(3):   # the host side of a batch: header parsing + copy into pinned memory + H2D of the compressed bytes
            codec.decode_stage(files)
            codec.sync()
        stage_ms = (time.perf_counter() - t0) / 3 * 1e3
    if mode == "decode":
        units = batch * 4
        step = lambda: codec.decode_run_device(d_frames.ptr, fh * fw * 3, fw * 3)
        alg = sum(len(f) for f in files) // len(files) + fh * fw * 3
    elif mode == "encode":
        units = batch
        d_bev.upload(bev_images)
        step = lambda: codec.encode_run_device(d_bev.ptr, batch, bw, bh, bh * bw * 3, bw * 3)
        step()
        alg = sum(len(f) for f in codec.files()) // batch + bh * bw * 3
    else:
        units = batch
        ns = SB.BevGenerator.get_args()
        for k, v in cfg.items():
            setattr(ns, k, v)
        bev = SB.BevGenerator(blend=w["blend"], balance=w["balance"], rig=W.rig_s(), device=dev)
        d_bev.free()
        d_bev = _ffi.DeviceBuffer(batch * bh * bev.out_pitch * 3, dev)
        image = bh * bev.out_pitch * 3

        def step():   # one asynchronous chain: the streams are ordered by events, the host does not wait in between
            codec.decode_run_device(d_frames.ptr, fh * fw * 3, fw * 3)
            codec.engine_waits(bev._engine.h)
            bev.run_device(d_frames.ptr, batch, None, d_bev.ptr, out_bytes=batch * image)
            codec.wait_engine(bev._engine.h)
            codec.encode_run_device(d_bev.ptr, batch, bw, bh, image, bev.out_pitch * 3)
        step()
        alg = 4 * (sum(len(f) for f in files) // len(files)) + sum(len(f) for f in codec.files()) // batch
    for _ in range(warmup):
        step()
    codec.sync()
    d.barrier()
    t0 = time.perf_counter()
    for i in range(steps):
        codec.timer_mark(i)
        step()
    codec.timer_mark(steps)
    codec.sync()
    d.barrier()
    wall = d.max(time.perf_counter() - t0)
    ev_ms = d.max(codec.timer_between(0, steps))
    info = codec.decode_info() if mode != "encode" else {}
    sizes = codec.files() if mode != "decode" else []
    if mode == "pipeline" and host_api and not os.environ.get("BEVW_BENCH_NO_HOST_API"):   # (profiling passes count the kernels of the resident steps only)
        # the host API on the same data, nothing resident: host byte strings in, host byte strings out (PCIe and host staging included; never
        # `value`).  jpeg_stream = consecutive batches pipelined (a host thread stages batch i + 1 and this thread fetches batch i - 1 while the
        # GPU runs batch i); jpeg = one synchronous call per batch
        sets = [tuple(files[4 * b:4 * b + 4]) for b in range(batch)]
        bev.jpeg(sets)
        reps = max(3, steps // 3)
        t1 = time.perf_counter()
        for _ in range(reps):
            out_files = bev.jpeg(sets)
        call_s = (time.perf_counter() - t1) / reps
        list(bev.jpeg_stream([sets] * 3))
        nb = max(6, steps)
        t1 = time.perf_counter()
        for out_files in bev.jpeg_stream(sets for _ in range(nb)):
            pass
        stream_s = (time.perf_counter() - t1) / nb
        extra["host_api_frames_per_s"] = round(batch / stream_s)
        extra["host_api_ms_per_batch"] = round(stream_s * 1e3, 3)
        extra["host_api_unpipelined_frames_per_s"] = round(batch / call_s)
        extra["host_api_unpipelined_ms_per_batch"] = round(call_s * 1e3, 3)
        extra["host_api_note"] = ("BevGenerator.jpeg_stream (pipelined over %d batches) / BevGenerator.jpeg (one synchronous call per batch) on host byte "
                                  "strings, PCIe and host staging included (%.1f MB of files in, %.1f MB out per batch); never `value`" % (
                                      nb, sum(len(f) for f in files) / 1e6, sum(len(f) for f in out_files) / 1e6))
    if mode != "encode":
        extra.update(subsequences_per_image=info["subsequences"] // info["images"], fixed_point_rounds_max=info["rounds"],
                     huffman_table_sets=info["table_sets"])
    if sizes:
        extra["bytes_per_output_file"] = sum(len(f) for f in sizes) // len(sizes)
    if stage_ms is not None:
        extra["host_stage_ms_per_batch"] = round(stage_ms, 3)
        extra["host_stage_files_per_s"] = round(batch * 4 / (stage_ms * 1e-3))
        extra["host_stage_note"] = ("header parsing + copy into pinned memory (host threads) + H2D of the compressed bytes for one batch; outside the timed "
                                    "region (inputs resident = the files' entropy-coded bytes in HBM); jpeg_stream overlaps it with the kernels")
    cpu = None
    if d.rank == 0 and d.world == 1 and cpu_seconds > 0 and mode != "pipeline":
        cpu = jpeg_cpu_baseline(mode, uniq_files, list(uniq_bev), cpu_seconds)
    launch_ms = ev_ms / steps
    achieved = alg * units / (launch_ms * 1e-3) / 1e9
    hbm = {"achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "algorithmic_bytes_per_unit": alg}
    prof = jpeg_valu_profile(workload if a.jpeg_source != "repo" else workload + "_repo")
    if prof:
        ginst = prof["valu_wave_insts_per_unit"] * units / (launch_ms * 1e-3) / 1e9
        peak = prof.get("peak_ginst") or VALU_PEAK_GINST_FULL_RATE
        roof = {"bound": "valu_issue", "achieved": ginst, "peak": peak, "unit": "G wave64 VALU instructions/s", "frac": ginst / peak,
                "peak_basis": prof.get("peak_basis") or "1024 SIMDs x 2.4 GHz / 2 clocks: every instruction counted as full rate (no instruction mix in profiles/jpeg_valu.json)",
                "clk_per_inst": prof.get("clk_per_inst"), "frac_if_all_full_rate": ginst / VALU_PEAK_GINST_FULL_RATE,
                "traffic": None, "kernel_ms": launch_ms, "units_per_launch": units, "valu_wave_insts_per_unit": prof["valu_wave_insts_per_unit"],
                "source": prof.get("source"), "hbm": hbm,
                "note": "the entropy stages are bound by VALU issue slots (every lane walks its own bit stream), not by HBM: achieved = wave-level VALU "
                        "instructions of one step (static figure of the committed PMC pass) / the step's measured time; peak = the issue rate of the "
                        "kernels' own instruction mix (peak_basis)"}
    else:
        roof = dict({"bound": "hbm", "traffic": None, "kernel_ms": launch_ms, "units_per_launch": units,
                     "note": "compressed bytes + pixels per unit; the entropy stages are VALU-issue bound, not HBM bound (DESIGN.md section 9); no "
                             "profiles/jpeg_valu.json entry for this workload"}, **hbm)
    out = {"metric": w["metric"], "value": units * d.world * steps / wall, "unit": w["unit"], "n_gpus": min(d.world, max(1, _ffi.device_count())),
           "ranks": d.world, "steps": steps, "warmup": warmup, "ms_per_step": wall / steps * 1e3, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "u8", "data": "synthetic",
           "config": dict({"workload": workload, "batch_per_gpu": batch, "units_per_step": units, "device": _ffi.device_name(dev),
                           "sharding": "files across ranks, no data-path collective"}, **extra),
           "roofline": roof, "cpu_baseline": cpu}
    codec.close()
    d_frames.free()
    d_bev.free()
    return out


def main_jpeg(a, d, w, dev):
    out = jpeg_measure(a, d, w, dev, a.workload, a.steps, a.warmup, 0.0 if a.no_cpu_baseline else min(a.cpu_seconds, 8.0))
    if d.rank == 0:
        print(json.dumps(out), flush=True)
    d.close()


def f4_summary(a, d, dev):
    """The JPEG wire format either side of the path in the DEFAULT run, so that the driver's bench record carries it:
    decode / encode / files-in-file-out rates with their roofline and the libjpeg-turbo CPU baseline.  Each leg is this script run on its JPEG
    workload in a process of its own (about 8 s each): HIP multiplexes a process's streams onto a few hardware queues, and next to the streams
    the stitch measurement has left behind the decoder's two slices landed on one queue and ran one after the other (3.7 instead of 2.8 ms)."""
    import subprocess

    res = {}
    for name, cpu_s in (("jpeg_decode_b64", 2.5), ("jpeg_encode_b64", 2.0), ("jpeg_bev_jpeg_b64", 0.0)):
        cmd = [sys.executable, os.path.abspath(__file__), "--workload", name, "--steps", "12", "--warmup", "4", "--cpu-seconds", str(cpu_s)]
        if cpu_s <= 0:
            cmd.append("--no-cpu-baseline")
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
        try:
            o = json.loads(r.stdout.strip().splitlines()[-1])
        except (IndexError, ValueError):
            res[name] = {"error": (r.stderr or r.stdout)[-400:]}
            continue
        keep = {"metric": o["metric"], "value": o["value"], "unit": o["unit"], "ms_per_step": o["ms_per_step"], "steps": o["steps"],
                "units_per_step": o["config"]["units_per_step"],
                "roofline": {k: o["roofline"].get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "kernel_ms", "peak_basis", "clk_per_inst")},
                "cpu_baseline": o["cpu_baseline"] and {k: o["cpu_baseline"].get(k) for k in ("value", "unit", "cores", "kind", "sample")}}
        for k in ("fixed_point_rounds_max", "host_api_frames_per_s", "host_api_ms_per_batch", "host_api_unpipelined_frames_per_s", "host_stage_ms_per_batch",
                  "bytes_per_output_file"):
            if k in o["config"]:
                keep[k] = o["config"][k]
        res[name] = keep
    res["note"] = ("reference wire format: cv2.imread x 4 (main.py:74-77), cv2.imwrite (surroundBEV.py:340); inputs resident = compressed bytes / device images; "
                   "each leg = `bench.py --workload <name> --steps 12 --warmup 4` in its own process")
    return res


def main():
    a = parse_args()
    d = Dist()
    if d.world != a.gpus and d.world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={d.world}")
    from cameracalibration_amd import _ffi, workloads as W

    _ffi.prefer_hw_queues(8)
    _ffi.require_device()  # loud: this bench has no CPU path
    dev = d.local_rank if _ffi.device_count() > d.local_rank else 0
    w = WORKLOADS[a.workload]
    for k, v in w.get("env", {}).items():   # switches the library reads once per process, before its first use
        os.environ[k] = v
    if w["kind"] == "jpeg":
        return main_jpeg(a, d, w, dev)
    batch = a.batch or w["batch"]
    sched = {"auto": _ffi.SCHED_AUTO, "pixel": _ffi.SCHED_PER_PIXEL, "plan": _ffi.SCHED_TILE_PLAN}[a.schedule]
    alg_bytes = W.ALGORITHMIC_BYTES[a.workload]
    cpu = None

    units_world = d.world   # ranks that each contribute `batch` units per step
    other_layout = None     # (name, harness factory) of the second device-image layout of the BEV workloads
    if w["kind"] == "camera":
        cfg, rig = W.CONFIG_4K, W.rig_4k()
        from cameracalibration_amd.SurroundBirdEyeView import cameraShard as CS, surroundBEV as SB

        ns = SB.BevGenerator.get_args()
        for k, v in cfg.items():
            setattr(ns, k, v)
        t_build = time.perf_counter()
        # the exchange is libbevwarp's native RCCL layer (csrc/bevw_comm.h); torch.distributed here is only the bench's barrier
        gen = CS.CameraShardedBev(w["blend"], w["balance"], rig=rig, rank=d.rank, world_size=d.world, device=dev)
        t_build = time.perf_counter() - t_build
        fw, fh, bw, bh = cfg["FRAME_WIDTH"], cfg["FRAME_HEIGHT"], cfg["BEV_WIDTH"], cfg["BEV_HEIGHT"]
        unique = W.synthetic_frames(a.unique_sets, fw, fh, seed=W.SEED + gen.group)
        mine = np.ascontiguousarray(unique[:, list(gen.cams)])
        d_in = _ffi.DeviceBuffer(batch * mine[0].nbytes, dev)
        upload_replicated(d_in, mine, batch)
        pipe = CS.ResidentShardPipeline(gen, batch)
        make_buffers = lambda: ((), (lambda: pipe.step(d_in.ptr, None)))   # the pipeline owns its buffers: one placement
        e = gen.engine
        sync = e.sync
        tstart = lambda: _ffi.check(_ffi.lib().bevw_timer_start(e.h))

        def tstop():
            import ctypes as C
            ms = C.c_float()
            _ffi.check(_ffi.lib().bevw_timer_stop(e.h, C.byref(ms)))
            return float(ms.value)
        tmark = lambda i: _ffi.check(_ffi.lib().bevw_timer_mark(e.h, i))

        def tbetween(a_, b_):
            import ctypes as C
            ms = C.c_float()
            _ffi.check(_ffi.lib().bevw_timer_between(e.h, a_, b_, C.byref(ms)))
            return float(ms.value)
        units_world = max(1, d.world // 4)
        parity = None
        if d.world > 1:
            # Before anything is timed on more than one GPU: the same ranks, the same native RCCL exchange, two seeded frame sets through the
            # host-array call, and every group's stitch rank compares with BevGenerator on its own GPU -- bit-exact or the run stops.  (The rank > 0
            # branches of bevw_shard_gather_parts / bevw_shard_allgather_vsums can only execute on a multi-GPU box: tests/test_camera_shard.py::
            # test_rccl_world_n_parity is the same check under pytest.)
            rng = np.random.default_rng(W.SEED + 1000 + gen.group)
            pf = rng.integers(0, 256, (2, 4, fh, fw, 3), dtype=np.uint8)
            got = gen(np.ascontiguousarray(pf[:, list(gen.cams)]), None, root=gen.ranks[0])
            bad = 0.0
            if got is not None:
                ref = SB.BevGenerator(blend=w["blend"], balance=w["balance"], rig=rig, device=dev)
                bad = 0.0 if np.array_equal(got, ref.batch(pf)) else 1.0
                del ref
            if d.max(bad) > 0:
                raise SystemExit("camera-shard parity check FAILED: the RCCL exchange does not reproduce BevGenerator's bytes")
            parity = "passed: 2 frame sets per camera group over the native RCCL exchange == BevGenerator on the stitch rank's GPU, bit-exact"
        extra = {"frame": [fw, fh], "bev": [bw, bh], "blend": w["blend"], "balance": w["balance"], "schedule": "tile_plan", "parity_check": parity,
                 "table_build_s": round(t_build, 3), "cameras_per_rank": len(gen.cams), "camera_groups": units_world,
                 "part_boxes": [list(b) for b in gen.boxes],
                 "transport": "single rank" if d.world == 1 else "rccl (native: ncclAllGather + grouped ncclSend/ncclRecv on the engine stream)"}
        if d.rank == 0 and d.world == 1 and not a.no_cpu_baseline:
            cpu = cpu_baseline_bev(w, cfg, rig, unique, a.cpu_seconds)
    elif w["kind"] == "bev":
        cfg = {"S": W.CONFIG_S, "4K": W.CONFIG_4K}[w["cfg"]]
        rig = {"S": W.rig_s, "4K": W.rig_4k}[w["cfg"]]()
        from cameracalibration_amd.SurroundBirdEyeView import surroundBEV as SB

        ns = SB.BevGenerator.get_args()
        for k, v in cfg.items():
            setattr(ns, k, v)
        fw, fh, bw, bh = cfg["FRAME_WIDTH"], cfg["FRAME_HEIGHT"], cfg["BEV_WIDTH"], cfg["BEV_HEIGHT"]
        unique = W.synthetic_frames(a.unique_sets, fw, fh, seed=W.SEED + d.rank)
        proj = w.get("projection", "lut")
        layouts = [a.output_pitch, "dense" if a.output_pitch != "dense" else "aligned"] if proj == "lut" else ["dense"]   # (a number = pixels per row, experiments)

        def engine(layout):
            t0 = time.perf_counter()
            g = SB.BevGenerator(blend=w["blend"], balance=w["balance"], rig=rig, device=dev, schedule=sched, projection=proj,
                                output_pitch=layout)
            return g, time.perf_counter() - t0

        def harness(g):
            def make_buffers():
                b_in = _ffi.DeviceBuffer(batch * unique[0].nbytes, dev)
                b_out = _ffi.DeviceBuffer(batch * bh * g.out_pitch * 3, dev)
                upload_replicated(b_in, unique, batch)
                return (b_in, b_out), (lambda: g.run_device(b_in.ptr, batch, None, b_out.ptr, out_bytes=batch * g.out_image_bytes))
            return make_buffers, g.sync, g.timer_start, g.timer_stop, g.timer_mark, g.timer_between
        bev, t_build = engine(layouts[0])
        make_buffers, sync, tstart, tstop, tmark, tbetween = harness(bev)
        if len(layouts) > 1 and not a.single_layout:
            other_layout = (layouts[1], lambda: harness(engine(layouts[1])[0]))
        info = bev.plan_info()
        extra = {"frame": [fw, fh], "bev": [bw, bh], "blend": w["blend"], "balance": w["balance"],
                 "schedule": {1: "per_pixel", 2: "tile_plan"}[info["schedule"]], "table_build_s": round(t_build, 3),
                 "tiles": {"staged": info["tiles_staged"], "gather": info["tiles_gather"], "border": info["tiles_border"]},
                 "output_layout": ("device images [B][%d][%d][3] u8: rows padded to whole 64-byte sectors (bevw_set_output_pitch; host entry "
                                   "points return dense arrays, rows compacted inside the D2H copy)" % (bh, bev.out_pitch)) if bev.out_pitch != bw
                 else "device images [B][%d][%d][3] u8, dense (the reference's host layout)" % (bh, bw)}
        if d.rank == 0 and d.world == 1 and not a.no_cpu_baseline:
            cpu = cpu_baseline_bev(w, cfg, rig, unique, a.cpu_seconds)
    else:
        import ctypes as C
        ucfg = W.CONFIG_UNDISTORT
        K, D = W.undistort_calibration()
        fw, fh = ucfg["FRAME_WIDTH"], ucfg["FRAME_HEIGHT"]
        r = C.c_void_p()
        _ffi.check(_ffi.lib().bevw_fisheye_remapper_create(dev, fw, fh, _ffi.ptr(_ffi.f64(K, 9)), _ffi.ptr(_ffi.f64(D, 4)),
                                                           ucfg["FOCAL_SCALE"], ucfg["SIZE_SCALE"], 0.0, 0.0, C.byref(r)))
        unique = W.synthetic_frames(max(1, a.unique_sets // 2), fw, fh, seed=W.SEED + d.rank)
        imgs = unique.reshape(-1, fh, fw, 3)
        L = _ffi.lib()

        def make_buffers():
            b_in = _ffi.DeviceBuffer(batch * imgs[0].nbytes, dev)
            b_out = _ffi.DeviceBuffer(batch * imgs[0].nbytes, dev)
            upload_replicated(b_in, imgs, batch)
            return (b_in, b_out), (lambda: _ffi.check(L.bevw_remap_device(r, b_in.ptr, batch, b_out.ptr)))
        sync = lambda: _ffi.check(L.bevw_remapper_sync(r))
        tstart = lambda: _ffi.check(L.bevw_remapper_timer_start(r))

        def tstop():
            ms = C.c_float()
            _ffi.check(L.bevw_remapper_timer_stop(r, C.byref(ms)))
            return float(ms.value)
        tmark = lambda i: _ffi.check(L.bevw_remapper_timer_mark(r, i))

        def tbetween(a_, b_):
            ms = C.c_float()
            _ffi.check(L.bevw_remapper_timer_between(r, a_, b_, C.byref(ms)))
            return float(ms.value)
        # (the remapper compiles its maps into a 1-camera unit plan; k_remap_lut, one thread per pixel, only when that plan is unusable)
        extra = {"frame": [fw, fh], "schedule": "tile_plan (1-camera plan of the undistort maps)"}
        if d.rank == 0 and d.world == 1 and not a.no_cpu_baseline:
            cpu = cpu_baseline_undistort(w, K, D, ucfg, unique, a.cpu_seconds)

    # Placements.  The step time of these request-bound kernels follows the DRAM latency the physical pages of the OUTPUT buffer
    # happen to give (profiles/r03/placement.md: 0.47 .. 0.58 ms for config 3 from the same build and the same process, offsets inside
    # an allocation do not matter, the channels are evenly loaded in fast and slow draws).  One allocation is one random draw, so the
    # buffers are freed and re-allocated `placements` times (a growing dummy allocation shifts the heap in between), W warm-up and
    # exactly K timed steps run on each, and the MEDIAN placement is what the line reports; all draws are listed next to it.
    placements = a.placements or (1 if w["kind"] == "camera" else 5)

    def measure(make_buffers, sync, tstart, tstop, tmark, tbetween):
        draws, dummies = [], []
        for pi in range(placements):
            bufs, step = make_buffers()
            for _ in range(a.warmup):
                step()
            sync()
            d.barrier()
            t0 = time.perf_counter()
            tstart()
            for i in range(a.steps):
                tmark(i)     # an event in front of every step, recorded on the engine's stream without synchronising
                step()
            tmark(a.steps)
            ev_ms = tstop()  # records the stop event on the engine's stream and waits for it
            sync()
            d.barrier()
            wall = time.perf_counter() - t0
            wall = d.max(wall)
            ev_ms = d.max(ev_ms)
            laps = sorted(tbetween(i, i + 1) for i in range(a.steps))
            lap_median = d.max(laps[len(laps) // 2] if len(laps) % 2 else 0.5 * (laps[len(laps) // 2 - 1] + laps[len(laps) // 2]))
            draws.append({"wall": wall, "ev_ms": ev_ms, "lap_median": lap_median, "lap_min": laps[0], "lap_max": laps[-1]})
            for b in bufs:
                b.free()
            if pi + 1 < placements and bufs:
                dummies.append(_ffi.DeviceBuffer((pi + 1) * 37 * 1024 * 1024 + 4096, dev))
        for b in dummies:
            b.free()
        order = sorted(range(placements), key=lambda i: draws[i]["wall"])
        mid = draws[order[placements // 2]] if placements % 2 else draws[order[placements // 2 - 1]]   # the (lower) median placement
        return mid, draws

    mid, draws = measure(make_buffers, sync, tstart, tstop, tmark, tbetween)
    wall, ev_ms, lap_median = mid["wall"], mid["ev_ms"], mid["lap_median"]
    laps = [mid["lap_min"], mid["lap_max"]]
    other = None
    if other_layout is not None:
        o_mid, o_draws = measure(*other_layout[1]())
        other = {"output_layout": other_layout[0], "ms_per_step": o_mid["wall"] / a.steps * 1e3, "kernel_ms_median": o_mid["lap_median"],
                 "value": batch * units_world * a.steps / o_mid["wall"],
                 "frac": alg_bytes * batch * a.steps / (o_mid["ev_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                 "placements_ms_per_step": [round(x["wall"] / a.steps * 1e3, 4) for x in o_draws]}

    # the measured yardstick beside the specification peak: what a plain copy kernel moves (bytes read + written per second)
    # between two fresh 1 GiB buffers on THIS box, default-policy and streaming accesses, the better of the two
    copy_gbs = None
    under_profiler = "rocprofiler" in os.environ.get("LD_PRELOAD", "") or bool(os.environ.get("ROCP_TOOL_LIBRARIES"))   # (keep profiles of the step clean)
    if d.rank == 0 and d.world == 1 and not os.environ.get("BEVW_BENCH_CHILD") and not under_profiler and not a.no_live_traffic:
        try:
            copy_gbs = max(_ffi.device_copy_rate(1 << 30, 10, st, dev) for st in (False, True) for _ in range(3))   # (fresh buffers per trial: placements differ)
        except Exception:
            copy_gbs = None
    lt = live_traffic(a, batch) if (d.rank == 0 and d.world == 1 and w["kind"] != "camera") else None
    traffic, traffic_fetch, traffic_write, traffic_source = lt if lt else static_traffic(a.workload, batch)
    agg = aggregate(units_world, batch, a.steps, wall, ev_ms, alg_bytes)
    value, launch_ms, achieved = agg["value"], agg["launch_ms"], agg["achieved_gbs"]
    out = {
        # n_gpus = distinct devices the ranks ran on (ranks sharing one GPU -- the gloo plumbing check on a 1-GPU box -- are not GPUs)
        "metric": w["metric"], "value": value, "unit": w["unit"], "n_gpus": min(d.world, max(1, _ffi.device_count())), "ranks": d.world,
        "steps": a.steps,
        "warmup": a.warmup, "ms_per_step": wall / a.steps * 1e3, "higher_is_better": True,
        "scaling": "strong" if (w["kind"] == "camera" and d.world <= 4) else "weak",
        "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": dict({"workload": a.workload, "batch_per_gpu": batch, "global_batch": batch * units_world,
                        "sharding": ("cameras across ranks, one send/recv of mask boxes per step" if w["kind"] == "camera"
                                     else "frames across ranks, no data-path collective"), "device": _ffi.device_name(dev),
                        "unique_frame_sets": int(a.unique_sets)}, **extra),
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "frac_best_placement": alg_bytes * batch * a.steps / (min(x["ev_ms"] for x in draws) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "frac_worst_placement": alg_bytes * batch * a.steps / (max(x["ev_ms"] for x in draws) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "traffic": traffic, "traffic_fetch": traffic_fetch, "traffic_write": traffic_write, "traffic_source": traffic_source,
                     "traffic_over_algorithmic": (traffic / (alg_bytes * batch)) if traffic else None,
                     # bytes the step really moves per second, against what a plain copy kernel moves on this box (read + write)
                     "moved_gbs": (traffic / (launch_ms * 1e-3) / 1e9) if traffic else None, "device_copy_gbs": copy_gbs,
                     "moved_over_device_copy": (traffic / (launch_ms * 1e-3) / 1e9 / copy_gbs) if (traffic and copy_gbs) else None,
                     "frac_of_device_copy": (achieved / copy_gbs) if copy_gbs else None,
                     "kernel_ms": launch_ms, "kernel_ms_median": lap_median, "kernel_ms_min": laps[0], "kernel_ms_max": laps[-1],
                     "frac_median": alg_bytes * batch / (lap_median * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "algorithmic_bytes_per_unit": alg_bytes, "units_per_launch": batch,
                     # the same step priced at 64-byte sector granularity (workloads.SECTOR_GRANULAR_BYTES): what a minifying map must move at least
                     "sector_floor_bytes_per_unit": W.SECTOR_GRANULAR_BYTES.get(a.workload),
                     "frac_sector_floor": (W.SECTOR_GRANULAR_BYTES[a.workload] * batch / (launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if a.workload in W.SECTOR_GRANULAR_BYTES else None},
        "placements": {"n": placements, "reported": "median placement (by wall time of its K steps)",
                       "ms_per_step": [round(x["wall"] / a.steps * 1e3, 4) for x in draws],
                       "kernel_ms_median": [round(x["lap_median"], 4) for x in draws]},
        "other_output_layout": other,
        "cpu_baseline": cpu,
    }
    # The reference's own output contract is a DENSE uint8[BH, BW, 3] image (surroundBEV.py:312-325): the figures on that device layout as
    # first-class fields, whichever layout the headline was measured on.  Rows of the undistort workload and of every
    # BEV width that is a multiple of 64 / 3 pixels are whole sectors already: dense IS the headline layout there.
    headline_dense = w["kind"] != "bev" or bev.out_pitch == bw
    if headline_dense:
        out["value_dense"], out["frac_dense"], out["ms_per_step_dense"] = value, achieved / HBM_PEAK_GBS, wall / a.steps * 1e3
    elif other is not None and other["output_layout"] == "dense":
        out["value_dense"], out["frac_dense"], out["ms_per_step_dense"] = other["value"], other["frac"], other["ms_per_step"]
    else:
        out["value_dense"] = out["frac_dense"] = out["ms_per_step_dense"] = None   # (--single-layout: not measured in this run)
    if a.workload == "direct_stitch_b256" and d.world == 1 and not a.no_f4 and not a.single_layout:
        try:
            out["f4"] = f4_summary(a, d, dev)
        except Exception as e:   # the headline must not depend on the JPEG legs
            out["f4"] = {"error": repr(e)}
    if d.rank == 0:
        print(json.dumps(out), flush=True)
    d.close()


if __name__ == "__main__":
    main()
