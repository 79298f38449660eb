"""Analytic projection mode (bevw_set_projection(BEVW_PROJ_ANALYTIC); SURVEY.md 8 row g1, BASELINE north star).

The reference's per-frame path is table-driven; this mode evaluates inverse homography + fisheye model per frame and pixel.
It is NOT the reference's arithmetic, so it is held to two different bars:
  * CPU: the fp64 NumPy specification (oracle/np_analytic.py) against the table path of the oracle on the reference's own
    frames -- the two must show the same picture (PSNR), which validates the projection formulas against the reference's
    tables; the differences are the tables' 1/32-pixel double quantisation (the LUT quirk), reported, not hidden.
  * GPU: the HIP kernel against that specification -- same formulas in fp64, so equal up to libm's atan / rounding ties:
    >= 99.9 % of the bytes identical, never more than 1 LSB apart.
"""
import numpy as np
import pytest

from cameracalibration_amd import workloads as W


@pytest.fixture(scope="module")
def ffi():
    from cameracalibration_amd import _ffi

    _ffi.require_device()
    return _ffi


@pytest.fixture(scope="module")
def SB():
    from cameracalibration_amd.SurroundBirdEyeView import surroundBEV

    return surroundBEV


def psnr(a, b, sel=None):
    d = (a.astype(np.float64) - b.astype(np.float64)) ** 2
    if sel is not None:
        d = d[sel]
    m = d.mean()
    return 99.0 if m == 0 else 10.0 * np.log10(255.0 ** 2 / m)


@pytest.mark.parametrize("blend", [False, True])
def test_specification_shows_the_same_picture_as_the_table_path(oracle, repo_rig, blend):
    from oracle import np_analytic

    cfg = dict(oracle.DEFAULT_CFG)
    frames = [repo_rig.image(n) for n in oracle.CAMERAS]
    ref = oracle.RefBevGenerator(repo_rig.rig, cfg, blend=blend, balance=False)(*frames)
    ana = np_analytic.AnalyticBevGenerator(repo_rig.rig, cfg, blend=blend)(*frames)
    assert ana.shape == ref.shape
    # pixels the table path fills from inside the undistorted image (outside it the tables hold the (0,0) quirk entries)
    inside = np.zeros(ref.shape[:2], bool)
    for i, n in enumerate(oracle.CAMERAS):
        _, _, valid = np_analytic.project(*repo_rig.rig[n], cfg)
        m = ref.shape and (oracle.direct_mask(n, cfg["BEV_WIDTH"], cfg["BEV_HEIGHT"], cfg["CAR_WIDTH"], cfg["CAR_HEIGHT"]) != 0)
        m = m if m.ndim == 2 else m[..., 0]
        inside |= valid & m
    p = psnr(ref, ana, inside)
    d = np.abs(ref.astype(np.int32) - ana.astype(np.int32))[inside]
    print("analytic vs table path, blend=%s: PSNR %.1f dB over %d pixels, mean |diff| %.2f, 99 %% <= %d LSB" % (
        blend, p, int(inside.sum()), d.mean(), int(np.percentile(d, 99))))
    assert inside.mean() > 0.5
    assert p > 38.0          # same picture (measured 43.7 / 44.2 dB); what is left is the tables' coordinate quantisation on the sample frames' edges
    assert d.mean() < 3.0


@pytest.mark.gpu
@pytest.mark.parametrize("blend", [False, True])
def test_hip_analytic_matches_the_specification(ffi, SB, oracle, repo_rig, blend):
    from oracle import np_analytic

    cfg = dict(oracle.DEFAULT_CFG)
    a = SB.BevGenerator.get_args()
    for k, v in cfg.items():
        setattr(a, k, v)
    frames = [repo_rig.image(n) for n in oracle.CAMERAS]
    car = np.zeros((cfg["BEV_HEIGHT"], cfg["BEV_WIDTH"], 3), np.uint8)
    car[300:700, 375:625] = 90
    bev = SB.BevGenerator(blend=blend, balance=False, rig=repo_rig.rig, projection='analytic')
    spec = np_analytic.AnalyticBevGenerator(repo_rig.rig, cfg, blend=blend)
    for c in (None, car):
        got, want = bev(*frames, c), spec(*frames, c)
        d = np.abs(got.astype(np.int32) - want.astype(np.int32))
        assert d.max() <= 1, int(d.max())
        assert (d == 0).mean() >= 0.999, float((d == 0).mean())
    # random frames, batch call
    rnd = W.synthetic_frames(2, cfg["FRAME_WIDTH"], cfg["FRAME_HEIGHT"], seed=5, kind="random")
    got = bev.batch(rnd)
    for b in range(2):
        d = np.abs(got[b].astype(np.int32) - spec(*rnd[b]).astype(np.int32))
        assert d.max() <= 1 and (d == 0).mean() >= 0.999


@pytest.mark.gpu
def test_hip_analytic_with_balance_runs_and_stays_close_to_the_table_path(ffi, SB, oracle, repo_rig):
    """blend + balance through the analytic kernel (per-tap luminance shift, channel sums, gain): no fp64 specification of the whole
    chain here -- the picture must stay the table path's (PSNR), the mode switch must be reversible in a fresh generator."""
    cfg = dict(oracle.DEFAULT_CFG)
    a = SB.BevGenerator.get_args()
    for k, v in cfg.items():
        setattr(a, k, v)
    frames = [repo_rig.image(n) for n in oracle.CAMERAS]
    lut = SB.BevGenerator(blend=True, balance=True, rig=repo_rig.rig)(*frames)
    ana = SB.BevGenerator(blend=True, balance=True, rig=repo_rig.rig, projection='analytic')(*frames)
    assert psnr(lut, ana) > 28.0
    with pytest.raises(Exception, match="projection should be lut/analytic"):
        SB.BevGenerator(rig=repo_rig.rig, projection='exact')


@pytest.mark.gpu
@pytest.mark.parametrize("frames_per_thread", [1, 32])
def test_hip_analytic_per_pixel_kernel_f32_against_the_specification(ffi, frames_per_thread):
    """north_star's wording taken literally (bench.py: direct_stitch_analytic_perpixel_b64): k_stitch_analytic on every pixel -- no unit plan
    (BEVW_ANALYTIC_UNITS=0), the inverse homography + fisheye model in fp32 evaluated per output pixel and per frame (BEVW_ANALYTIC_FRAMES=1; 32 =
    the kernel's default amortisation) -- against the fp64 specification oracle/np_analytic.py on the reference's own frames, batch of 3 with a
    sprite: PSNR, share of identical bytes and the maximum difference (an fp32 arithmetic of its own: not bit for bit).  The switches are read
    once per process, hence the child."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    code = (
        "import numpy as np, sys\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "from oracle import oracle as O, np_analytic as NA\n"
        "from cameracalibration_amd.SurroundBirdEyeView import surroundBEV as SB\n"
        "import test_analytic as T, conftest as CT\n"
        "rr = CT.RepoRig()\n"
        "a = SB.BevGenerator.get_args()\n"
        "for k, v in dict(O.DEFAULT_CFG).items(): setattr(a, k, v)\n"
        "frames = [rr.image(n) for n in O.CAMERAS]\n"
        "for blend in (False, True):\n"
        "    spec = NA.AnalyticBevGenerator(rr.rig, dict(O.DEFAULT_CFG), blend=blend)(*frames)\n"
        "    bev = SB.BevGenerator(blend=blend, rig=rr.rig, projection='analytic_f32')\n"
        "    got = bev.batch(np.stack([np.stack(frames)] * 3))\n"
        "    assert np.array_equal(got[0], got[1]) and np.array_equal(got[0], got[2])\n"
        "    d = np.abs(got[0].astype(np.int32) - spec.astype(np.int32))\n"
        "    print('per-pixel fp32 kernel vs fp64 specification, blend=%%s: PSNR %%.1f dB, %%.2f %%%% identical, max %%d LSB' %% (blend, T.psnr(spec, got[0]), 100 * (d == 0).mean(), int(d.max())))\n"
        "    assert T.psnr(spec, got[0]) > 55.0 and (d == 0).mean() > 0.97 and d.max() <= 2, (T.psnr(spec, got[0]), (d == 0).mean(), d.max())\n"
        "print('per-pixel analytic ok')\n") % (ROOT, os.path.join(ROOT, "tests"))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, BEVW_ANALYTIC_UNITS="0", BEVW_ANALYTIC_FRAMES=str(frames_per_thread)),
                       capture_output=True, text=True, timeout=600)
    print(r.stdout)
    assert r.returncode == 0 and "per-pixel analytic ok" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("blend", [False, True])
def test_hip_analytic_f32_against_fp64(ffi, SB, oracle, repo_rig, blend):
    """fp32 projection (positions good to ~1e-4 pixel) against the fp64 mode: judged by PSNR and max difference, not byte for byte"""
    cfg = dict(oracle.DEFAULT_CFG)
    a = SB.BevGenerator.get_args()
    for k, v in cfg.items():
        setattr(a, k, v)
    frames = [repo_rig.image(n) for n in oracle.CAMERAS]
    f64 = SB.BevGenerator(blend=blend, rig=repo_rig.rig, projection='analytic')(*frames)
    f32 = SB.BevGenerator(blend=blend, rig=repo_rig.rig, projection='analytic_f32')(*frames)
    d = np.abs(f64.astype(np.int32) - f32.astype(np.int32))
    print("analytic fp32 vs fp64, blend=%s: PSNR %.1f dB, %.2f %% identical, max %d LSB" % (blend, psnr(f64, f32), 100 * (d == 0).mean(), int(d.max())))
    assert psnr(f64, f32) > 55.0
    assert (d == 0).mean() > 0.97


@pytest.mark.gpu
@pytest.mark.parametrize("blend", [False, True])
def test_hip_analytic_units_on_the_bench_rig_batch_of_64(ffi, SB, oracle, blend):
    """The analytic mode on the unit schedule (wide plan: csrc/bevwarp.hip analytic_units_build, csrc/bevw_unit.h k_plan_unit_wide) on the bench
    rig with the bench's batch size, device-resident, all-random frames and a car sprite: every BEV of the batch is independent of its position
    (chunk / XCD mapping, the ragged tail of a 19-frame batch) and within 1 LSB of the fp64 specification, >= 99.9 % of the bytes identical."""
    from oracle import np_analytic

    cfg, rig = W.CONFIG_S, W.rig_s()
    a = SB.BevGenerator.get_args()
    for k, v in cfg.items():
        setattr(a, k, v)
    bw, bh = cfg["BEV_WIDTH"], cfg["BEV_HEIGHT"]
    uniq, batch = 2, 64
    frames = W.synthetic_frames(uniq, cfg["FRAME_WIDTH"], cfg["FRAME_HEIGHT"], seed=11, kind="random")
    rng = np.random.default_rng(5)
    car = np.zeros((bh, bw, 3), np.uint8)
    car[380:720, 400:680] = rng.integers(0, 256, (340, 280, 3), dtype=np.uint8)
    bev = SB.BevGenerator(blend=blend, balance=False, rig=rig, projection='analytic')
    spec = np_analytic.AnalyticBevGenerator(rig, cfg, blend=blend)
    d_in = ffi.DeviceBuffer(batch * frames[0].nbytes)
    d_car = ffi.DeviceBuffer(car.nbytes)
    d_out = ffi.DeviceBuffer(batch * bw * bh * 3)
    d_out.fill(0x5A)
    d_car.upload(car)
    for b in range(batch):
        d_in.upload(frames[b % uniq], offset=b * frames[0].nbytes)
    bev.run_device(d_in.ptr, batch, d_car.ptr, d_out.ptr)
    bev.sync()
    out = d_out.download((batch, bh, bw, 3))
    for b in range(uniq, batch):
        assert np.array_equal(out[b], out[b % uniq]), b
    for b in range(uniq):
        d = np.abs(out[b].astype(np.int32) - spec(*frames[b], car).astype(np.int32))
        assert d.max() <= 1, int(d.max())
        assert (d == 0).mean() >= 0.999, float((d == 0).mean())
    # a ragged batch through the host-buffer entry point, no sprite
    rag = bev.batch(np.concatenate([frames] * 9 + [frames[:1]]))
    want0 = spec(*frames[0])
    d = np.abs(rag[18].astype(np.int32) - want0.astype(np.int32))
    assert d.max() <= 1 and (d == 0).mean() >= 0.999
    assert np.array_equal(rag[18], rag[0]) and np.array_equal(rag[17], rag[1])
