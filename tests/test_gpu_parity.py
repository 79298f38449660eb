"""GPU parity: the HIP engine (through the C-ABI / the reference-API mirror) against the CPU oracle on the same seeded
inputs.  Integer / byte / index work -> the bar is BIT-EXACT (tolerance 0), which is inside BASELINE.json's
"+-1 LSB per channel" for the images.  Run with `-m gpu` on an MI355X.
"""
import ctypes as C

import numpy as np
import pytest

from cameracalibration_amd import workloads as W

pytestmark = pytest.mark.gpu

CAMS = W.CAMERA_NAMES


@pytest.fixture(scope="module")
def ffi():
    from cameracalibration_amd import _ffi

    _ffi.require_device()
    return _ffi


@pytest.fixture(scope="module")
def SB():
    from cameracalibration_amd.SurroundBirdEyeView import surroundBEV

    return surroundBEV


def set_args(SB, cfg):
    ns = SB.BevGenerator.get_args()
    for k, v in cfg.items():
        setattr(ns, k, v)


def make_pair(SB, oracle, rig, cfg, blend, balance, schedule=0):
    set_args(SB, cfg)
    bev = SB.BevGenerator(blend=blend, balance=balance, rig=rig, schedule=schedule)
    ref = oracle.RefBevGenerator(rig, cfg, blend=blend, balance=balance)
    return bev, ref


def maxdiff(a, b):
    return int(np.abs(a.astype(np.int32) - b.astype(np.int32)).max())


# A small rig whose geometry exercises every path quickly: the repo rig scaled to 320x256 frames -> 250x250 BEV.
SMALL_CFG = dict(FRAME_WIDTH=320, FRAME_HEIGHT=256, BEV_WIDTH=248, BEV_HEIGHT=250, CAR_WIDTH=62, CAR_HEIGHT=100,
                 FOCAL_SCALE=1.0, SIZE_SCALE=2.0)


def small_rig():
    out = {}
    A = np.diag([0.25, 0.25, 1.0])  # raw frame scaled by 1/4
    for n, (K, D, H) in W.repo_rig().items():
        Ks = A @ K
        # undistorted grid also scales by 1/4; BEV by 1/4: H_s = A . H . A^-1
        out[n] = (Ks, D.copy(), A @ H @ np.linalg.inv(A))
    return out


# ---------------------------------------------------------------------------------------------------------------
# tables
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("which", ["R", "S"])
def test_tables_bit_exact(ffi, SB, oracle, which):
    cfg, rig = (W.CONFIG_R, W.repo_rig()) if which == "R" else (W.CONFIG_S, W.rig_s())
    bev, ref = make_pair(SB, oracle, rig, cfg, blend=False, balance=False)
    for i, n in enumerate(CAMS):
        u1, u2 = bev.cameras[i].undistort_maps
        assert np.array_equal(u1, ref.cameras[i].undistort_maps[0]), f"{n} undistort map1"
        assert np.array_equal(u2, ref.cameras[i].undistort_maps[1]), f"{n} undistort map2"
        b1, b2 = bev.cameras[i].bev_maps
        assert np.array_equal(b1, ref.cameras[i].bev_maps[0]), f"{n} bev map1"
        assert np.array_equal(b2, ref.cameras[i].bev_maps[1]), f"{n} bev map2"
        assert np.array_equal(bev.masks[i].mask, ref.masks[i]), f"{n} direct mask"
    info = bev.plan_info()
    assert info["max_contributors"] <= 2 and info["plan_usable"]


@pytest.mark.parametrize("cfg_name", ["R", "R_main", "S", "small"])
def test_blend_masks_bit_exact(ffi, SB, oracle, cfg_name):
    cfg = {"R": W.CONFIG_R, "R_main": dict(W.CONFIG_R, CAR_WIDTH=200, CAR_HEIGHT=350), "S": W.CONFIG_S,
           "small": SMALL_CFG}[cfg_name]
    set_args(SB, cfg)
    ident = {n: (np.eye(3) * [100, 100, 1], np.zeros(4), np.eye(3)) for n in CAMS}
    bev = SB.BevGenerator(blend=True, balance=False, rig=ident)
    geo = (cfg["BEV_WIDTH"], cfg["BEV_HEIGHT"], cfg["CAR_WIDTH"], cfg["CAR_HEIGHT"])
    for i, n in enumerate(CAMS):
        want = oracle.blend_mask_for(n, *geo)
        got = bev.masks[i].mask
        assert np.array_equal(got, want), f"{n}: {np.count_nonzero(got != want)} px differ"
        assert np.array_equal(bev.masks[i].weight, oracle.blend_weight(want))


@pytest.mark.parametrize("geo", [(250, 251, 60, 100), (97, 64, 20, 30), (1000, 1000, 250, 400), (40, 40, 0, 0)])
def test_direct_masks_odd_sizes(ffi, SB, oracle, geo):
    cfg = dict(SMALL_CFG, BEV_WIDTH=geo[0], BEV_HEIGHT=geo[1], CAR_WIDTH=geo[2], CAR_HEIGHT=geo[3])
    set_args(SB, cfg)
    ident = {n: (np.eye(3) * [100, 100, 1], np.zeros(4), np.eye(3)) for n in CAMS}
    bev = SB.BevGenerator(blend=False, balance=False, rig=ident)
    for i, n in enumerate(CAMS):
        assert np.array_equal(bev.masks[i].mask, oracle.direct_mask(n, *geo)), n


@pytest.mark.parametrize("fillpoly,addweighted", [(0, 1), (1, 0), (0, 0)])
def test_compat_variants_bit_exact(ffi, SB, oracle, fillpoly, addweighted):
    """The OpenCV-version switches (bevw_set_compat / oracle.set_variant, tests/golden/README.md) move engine and oracle together:
    masks of the sample geometry and a blend + balance frame set stay bit-exact for every combination."""
    L = ffi.lib()
    try:
        ffi.check(L.bevw_set_compat(ffi.COMPAT_FILLPOLY, fillpoly))
        ffi.check(L.bevw_set_compat(ffi.COMPAT_ADDWEIGHTED, addweighted))
        oracle.set_variant(oracle.VARIANT_FILLPOLY, fillpoly)
        oracle.set_variant(oracle.VARIANT_ADDWEIGHTED, addweighted)
        assert L.bevw_get_compat(ffi.COMPAT_FILLPOLY) == fillpoly and L.bevw_get_compat(ffi.COMPAT_ADDWEIGHTED) == addweighted
        for blend in (False, True):
            bev, ref = make_pair(SB, oracle, W.repo_rig(), W.CONFIG_R, blend=blend, balance=False)
            for i, n in enumerate(CAMS):
                assert np.array_equal(bev.masks[i].mask, ref.masks[i]), f"{n} mask, blend={blend}"
        rig = small_rig()
        bev, ref = make_pair(SB, oracle, rig, SMALL_CFG, blend=True, balance=True)
        frames = W.synthetic_frames(3, SMALL_CFG["FRAME_WIDTH"], SMALL_CFG["FRAME_HEIGHT"], kind="random")
        got = bev.batch(frames)
        for b in range(frames.shape[0]):
            assert maxdiff(got[b], ref(*frames[b])) == 0
        # a handle keeps the values it was built with: flipping the process-wide defaults afterwards does not reach it
        ffi.check(L.bevw_set_compat(ffi.COMPAT_FILLPOLY, 1 - fillpoly))
        ffi.check(L.bevw_set_compat(ffi.COMPAT_ADDWEIGHTED, 1 - addweighted))
        assert np.array_equal(bev.batch(frames), got)
    finally:
        L.bevw_set_compat(ffi.COMPAT_FILLPOLY, 1)
        L.bevw_set_compat(ffi.COMPAT_ADDWEIGHTED, 1)
        oracle.set_variant(oracle.VARIANT_FILLPOLY, 1)
        oracle.set_variant(oracle.VARIANT_ADDWEIGHTED, 1)


@pytest.mark.parametrize("mode", [1, 3, 5, 9, 17, 33, 15, 29, 47])
def test_compat_warp_family_bit_exact(ffi, SB, oracle, mode):
    """BEVW_COMPAT_WARP / VARIANT_WARP: members of the float32 family (candidates for OpenCV >= 4.11's warpPerspective kernels) move engine and
    oracle together -- a3: the BEV look-up table (its 16UC1 half goes through the float kernel, the two-channel 16S half stays classic),
    a11: ExCalibrator.warp on an 8UC3 image, and a stitched frame set built on those tables.  Bit-exact for every member."""
    L = ffi.lib()
    try:
        ffi.check(L.bevw_set_compat(ffi.COMPAT_WARP, mode))
        oracle.set_variant(oracle.VARIANT_WARP, mode)
        assert L.bevw_get_compat(ffi.COMPAT_WARP) == mode
        rig = small_rig()
        bev, ref = make_pair(SB, oracle, rig, SMALL_CFG, blend=True, balance=False)
        for i, n in enumerate(CAMS):
            m1, m2 = bev.cameras[i].bev_maps
            assert np.array_equal(m1, ref.cameras[i].bev_maps[0]) and np.array_equal(m2, ref.cameras[i].bev_maps[1]), n
        oracle.set_variant(oracle.VARIANT_WARP, 0)
        classic = oracle.RefBevGenerator(rig, SMALL_CFG, blend=True, balance=False)
        oracle.set_variant(oracle.VARIANT_WARP, mode)
        assert any(not np.array_equal(ref.cameras[i].bev_maps[1], classic.cameras[i].bev_maps[1]) for i in range(4))   # the switch is not a no-op
        assert all(np.array_equal(ref.cameras[i].bev_maps[0], classic.cameras[i].bev_maps[0]) for i in range(4))       # the 16SC2 half is
        frames = W.synthetic_frames(2, SMALL_CFG["FRAME_WIDTH"], SMALL_CFG["FRAME_HEIGHT"], kind="random")
        got = bev.batch(frames)
        for b in range(2):
            assert maxdiff(got[b], ref(*frames[b])) == 0
        rng = np.random.default_rng(mode)
        img = rng.integers(0, 256, (120, 170, 3), dtype=np.uint8)
        H = np.array([[0.9, 0.12, -7.0], [-0.06, 1.1, 9.0], [2.0e-4, -1.0e-4, 1.0]])
        got = ffi_warp(ffi, img, H, (200, 140))
        assert np.array_equal(got, oracle.warp_perspective(img, H, (200, 140)))
        # a handle keeps the member it was built with
        ffi.check(L.bevw_set_compat(ffi.COMPAT_WARP, 0))
        assert maxdiff(bev.batch(frames)[0], ref(*frames[0])) == 0
    finally:
        L.bevw_set_compat(ffi.COMPAT_WARP, 0)
        oracle.set_variant(oracle.VARIANT_WARP, 0)
    assert L.bevw_set_compat(ffi.COMPAT_WARP, 2) != 0 and L.bevw_set_compat(ffi.COMPAT_WARP, 64) != 0   # even / out of range: refused


def ffi_warp(ffi, img, H, dsize):
    """cv2.warpPerspective(img, H, dsize) through the C-ABI (bevw_warp_perspective_u8c3: what ExCalibrator.warp calls)."""
    out = np.empty((dsize[1], dsize[0], 3), np.uint8)
    ffi.check(ffi.lib().bevw_warp_perspective_u8c3(0, ffi.ptr(np.ascontiguousarray(img)), img.shape[1], img.shape[0], ffi.ptr(ffi.f64(H, 9)),
                                                   dsize[0], dsize[1], 1, ffi.ptr(out)))
    return out


def test_compat_remap_tie_rule_bit_exact(ffi, SB, oracle):
    """BEVW_COMPAT_REMAP 1 / VARIANT_REMAP 1: the exact weighted sum rounded half to even instead of half up (what a float kernel ending in
    cvRound gives; a candidate for OpenCV >= 4.11's remap).  The engine runs the per-pixel schedule then; all four BevGenerator modes, the
    Camera methods and a stand-alone remapper stay bit-exact against the oracle, and the tile plan is refused."""
    L = ffi.lib()
    try:
        ffi.check(L.bevw_set_compat(ffi.COMPAT_REMAP, 1))
        oracle.set_variant(oracle.VARIANT_REMAP, 1)
        rig = small_rig()
        frames = W.synthetic_frames(2, SMALL_CFG["FRAME_WIDTH"], SMALL_CFG["FRAME_HEIGHT"], kind="random")
        seen_tie = False
        for blend, balance in [(False, False), (True, False), (False, True), (True, True)]:
            bev, ref = make_pair(SB, oracle, rig, SMALL_CFG, blend, balance)
            assert bev.plan_info()["schedule"] == 1 and bev.out_pitch == SMALL_CFG["BEV_WIDTH"]
            got = bev.batch(frames)
            for b in range(2):
                assert maxdiff(got[b], ref(*frames[b])) == 0, (blend, balance)
            if not blend and not balance:
                oracle.set_variant(oracle.VARIANT_REMAP, 0)
                up = oracle.RefBevGenerator(rig, SMALL_CFG, blend=False, balance=False)(*frames[0])
                oracle.set_variant(oracle.VARIANT_REMAP, 1)
                seen_tie = not np.array_equal(up, got[0])
                cam = bev.cameras[0]
                assert np.array_equal(cam.raw2bev(frames[0][0]), ref.cameras[0].raw2bev(frames[0][0]))
                assert np.array_equal(cam.undistort(frames[0][0]), ref.cameras[0].undistort(frames[0][0]))
        assert seen_tie   # random frames do hit exact ties: the rule changes some pixel
        with pytest.raises(Exception):
            SB.BevGenerator(rig=rig, schedule=ffi.SCHED_TILE_PLAN)
        import ctypes as C
        rng = np.random.default_rng(9)
        m1 = rng.integers(0, 60, (40, 48, 2)).astype(np.int16)
        m2 = rng.integers(0, 1024, (40, 48)).astype(np.uint16)
        img = rng.integers(0, 256, (3, 64, 64, 3), dtype=np.uint8)
        r = C.c_void_p()
        ffi.check(L.bevw_remapper_from_maps(0, 64, 64, ffi.ptr(m1), ffi.ptr(m2), 48, 40, C.byref(r)))
        out = np.empty((3, 40, 48, 3), np.uint8)
        ffi.check(L.bevw_remap(r, ffi.ptr(img), 3, ffi.ptr(out)))
        L.bevw_remapper_destroy(r)
        for b in range(3):
            assert np.array_equal(out[b], oracle.remap(img[b], m1, m2))
    finally:
        L.bevw_set_compat(ffi.COMPAT_REMAP, 0)
        oracle.set_variant(oracle.VARIANT_REMAP, 0)


# ---------------------------------------------------------------------------------------------------------------
# BevGenerator.__call__ on the reference's own sample data (config 1)
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("blend,balance", [(False, False), (True, False), (False, True), (True, True)])
@pytest.mark.parametrize("schedule", [1, 2])
def test_repo_data_bit_exact(ffi, SB, oracle, repo_rig, blend, balance, schedule):
    cfg = dict(W.CONFIG_R, CAR_WIDTH=200, CAR_HEIGHT=350) if blend else W.CONFIG_R  # main.py:80-81 for the blend demo
    bev, ref = make_pair(SB, oracle, repo_rig.rig, cfg, blend, balance, schedule)
    frames = repo_rig.frames()
    car = SB.padding(repo_rig.image("car"), cfg["BEV_WIDTH"], cfg["BEV_HEIGHT"])
    assert bev.plan_info()["schedule"] == schedule
    for c in (None, car):
        got, want = bev(*frames, c), ref(*frames, c)
        assert got.dtype == np.uint8 and got.shape == want.shape
        assert maxdiff(got, want) == 0, f"{np.count_nonzero(got != want)} bytes differ"


def test_camera_methods_bit_exact(ffi, SB, oracle, repo_rig):
    bev, ref = make_pair(SB, oracle, repo_rig.rig, W.CONFIG_R, False, False)
    frames = repo_rig.frames()
    for i in range(4):
        assert np.array_equal(bev.cameras[i].raw2bev(frames[i]), ref.cameras[i].raw2bev(frames[i]))
    und = bev.cameras[0].undistort(frames[0])
    want = ref.cameras[0].undistort(frames[0])
    assert np.array_equal(und, want)
    assert np.array_equal(bev.cameras[0].warp_homography(und), ref.cameras[0].warp_homography(want))


# ---------------------------------------------------------------------------------------------------------------
# seeded synthetic batches, both schedules, all modes (small rig -> seconds on the oracle)
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("blend,balance", [(False, False), (True, False), (False, True), (True, True)])
@pytest.mark.parametrize("kind", ["random", "smooth"])
def test_small_rig_batches(ffi, SB, oracle, blend, balance, kind):
    rig = small_rig()
    frames = W.synthetic_frames(5, SMALL_CFG["FRAME_WIDTH"], SMALL_CFG["FRAME_HEIGHT"], seed=7, kind=kind)
    rng = np.random.default_rng(3)
    car = np.zeros((SMALL_CFG["BEV_HEIGHT"], SMALL_CFG["BEV_WIDTH"], 3), np.uint8)
    car[60:190, 80:170] = rng.integers(0, 256, (130, 90, 3), dtype=np.uint8)
    outs = {}
    for schedule in (1, 2):
        bev, ref = make_pair(SB, oracle, rig, SMALL_CFG, blend, balance, schedule)
        assert bev.plan_info()["schedule"] == schedule
        got = bev.batch(frames, car)
        for b in range(frames.shape[0]):
            want = ref(*frames[b], car)
            assert maxdiff(got[b], want) == 0, (schedule, b, np.count_nonzero(got[b] != want))
        outs[schedule] = got
    assert np.array_equal(outs[1], outs[2])


@pytest.mark.parametrize("bw,blend,balance,with_car", [(250, True, False, False), (249, False, False, True), (251, True, True, True),
                                                        (250, False, True, False)])
def test_bev_width_not_a_multiple_of_4_stays_on_the_tile_plan(ffi, SB, oracle, bw, blend, balance, with_car):
    """no per-pixel cliff for bw % 4 != 0 -- the plan kernels write rows of a padded pitch and one
    compaction pass brings them to the caller's layout (Plan::pitch, k_plan_unpad)."""
    cfg = dict(SMALL_CFG, BEV_WIDTH=bw, BEV_HEIGHT=251)
    rig = small_rig()
    bev, ref = make_pair(SB, oracle, rig, cfg, blend, balance, 0)
    info = bev.plan_info()
    assert info["schedule"] == 2 and info["plan_usable"]
    frames = W.synthetic_frames(3, cfg["FRAME_WIDTH"], cfg["FRAME_HEIGHT"], seed=11, kind="random")
    car = None
    if with_car:
        car = np.zeros((251, bw, 3), np.uint8)
        x0, y0 = (bw - cfg["CAR_WIDTH"]) // 2, (251 - cfg["CAR_HEIGHT"]) // 2
        car[y0:y0 + cfg["CAR_HEIGHT"], x0:x0 + cfg["CAR_WIDTH"]] = np.random.default_rng(3).integers(0, 256, (cfg["CAR_HEIGHT"], cfg["CAR_WIDTH"], 3), dtype=np.uint8)
    got = bev.batch(frames, car)
    for b in range(3):
        assert maxdiff(got[b], ref(*frames[b], car)) == 0
    # and the explicit per-pixel schedule agrees
    bev1 = SB.BevGenerator(blend=blend, balance=balance, rig=rig, schedule=1)
    assert np.array_equal(bev1.batch(frames, car), got)


def test_frames_that_are_not_dword_multiples_fall_back_to_per_pixel(ffi, SB, oracle):
    """The plan's aligned 12-byte footprint reads need every frame of a set to start on a 4-byte boundary: 321 x 255 x 3 bytes
    does not, so AUTO picks the per-pixel schedule and an explicit schedule=2 is refused."""
    cfg = dict(SMALL_CFG, FRAME_WIDTH=321, FRAME_HEIGHT=255)
    rig = small_rig()
    bev, ref = make_pair(SB, oracle, rig, cfg, True, False, 0)
    assert bev.plan_info()["schedule"] == 1 and not bev.plan_info()["plan_usable"]
    frames = W.synthetic_frames(2, cfg["FRAME_WIDTH"], cfg["FRAME_HEIGHT"], seed=11, kind="random")
    got = bev.batch(frames)
    for b in range(2):
        assert maxdiff(got[b], ref(*frames[b])) == 0
    with pytest.raises(Exception, match="tile plan unusable"):
        SB.BevGenerator(blend=True, rig=rig, schedule=2)


def test_edge_cases(ffi, SB, oracle):
    rig = small_rig()
    bev, ref = make_pair(SB, oracle, rig, SMALL_CFG, False, False)
    fh, fw = SMALL_CFG["FRAME_HEIGHT"], SMALL_CFG["FRAME_WIDTH"]
    assert bev.batch(np.empty((0, 4, fh, fw, 3), np.uint8)).shape == (0, SMALL_CFG["BEV_HEIGHT"], SMALL_CFG["BEV_WIDTH"], 3)
    with pytest.raises(Exception, match="frames must be uint8"):
        bev.batch(np.zeros((1, 4, fh, fw + 1, 3), np.uint8))
    with pytest.raises(Exception, match="car must be padded"):
        bev.batch(np.zeros((1, 4, fh, fw, 3), np.uint8), np.zeros((3, 3, 3), np.uint8))
    # constant frames: every interior BEV pixel of a single-camera region reproduces the colour exactly
    frames = W.synthetic_frames(1, fw, fh, seed=5, kind="constant")
    got = bev.batch(frames)[0]
    assert maxdiff(got, ref(*frames[0])) == 0
    # maximum values saturate identically
    full = np.full((1, 4, fh, fw, 3), 255, np.uint8)
    for blend, balance in [(True, True), (False, False)]:
        b2, r2 = make_pair(SB, oracle, rig, SMALL_CFG, blend, balance)
        assert maxdiff(b2.batch(full)[0], r2(*full[0])) == 0
        zero = np.zeros_like(full)
        assert maxdiff(b2.batch(zero)[0], r2(*zero[0])) == 0  # balance with 0/0 gains


def test_lut_border_entries_are_exercised(ffi, SB, oracle):
    """A homography that pushes part of the BEV outside the undistort grid and the raw frame: partial footprints,
    whole-footprint misses and the (0,0)+code 0 'quirk' entries all appear, on both schedules."""
    rig = small_rig()
    shift = np.array([[1.0, 0, -90.0], [0, 1.0, 60.0], [0, 0, 1.0]])
    rig2 = {n: (K * [[1.0, 1, 0.55], [1, 1.0, 0.55], [1, 1, 1]], D, shift @ H) for n, (K, D, H) in rig.items()}
    frames = W.synthetic_frames(2, SMALL_CFG["FRAME_WIDTH"], SMALL_CFG["FRAME_HEIGHT"], seed=13, kind="random")
    for schedule in (1, 2):
        bev, ref = make_pair(SB, oracle, rig2, SMALL_CFG, True, True, schedule)
        lut1 = ref.cameras[0].bev_maps[0]
        assert (lut1[..., 0] < 0).any() or (lut1[..., 0] >= SMALL_CFG["FRAME_WIDTH"] - 1).any()
        got = bev.batch(frames)
        for b in range(2):
            assert maxdiff(got[b], ref(*frames[b])) == 0, schedule


# ---------------------------------------------------------------------------------------------------------------
# exported helpers, InCalibrator.undistort (config 2), ExCalibrator.warp
# ---------------------------------------------------------------------------------------------------------------
def test_balance_helpers(ffi, SB, oracle, repo_rig):
    frames = repo_rig.frames()
    got = SB.luminance_balance(frames)
    want = oracle.luminance_balance(frames)
    for g, w_ in zip(got, want):
        assert np.array_equal(g, w_)
    rng = np.random.default_rng(0)
    img = rng.integers(0, 200, (123, 77, 3), dtype=np.uint8)
    assert np.array_equal(SB.color_balance(img), oracle.color_balance(img))
    odd = [rng.integers(0, 256, (37, 53, 3), dtype=np.uint8) for _ in range(4)]  # frame bytes not a multiple of 16/48
    for g, w_ in zip(SB.luminance_balance(odd), oracle.luminance_balance(odd)):
        assert np.array_equal(g, w_)


def test_incalibrator_undistort_config2(ffi, oracle):
    from cameracalibration_amd.IntrinsicCalibration import InCalibrator

    ucfg = W.CONFIG_UNDISTORT
    a = InCalibrator.get_args()
    a.FRAME_WIDTH, a.FRAME_HEIGHT, a.FOCAL_SCALE, a.SIZE_SCALE = (ucfg["FRAME_WIDTH"], ucfg["FRAME_HEIGHT"],
                                                                  ucfg["FOCAL_SCALE"], ucfg["SIZE_SCALE"])
    K, D = W.undistort_calibration()
    cal = InCalibrator("fisheye")
    data = cal.set_calibration(K, D)
    Kd = oracle.camera_mat_dst(K, a.FRAME_WIDTH, a.FRAME_HEIGHT, a.FOCAL_SCALE, a.SIZE_SCALE)
    m1, m2 = oracle.fisheye_init_undistort_rectify_map(K, D, Kd, (a.FRAME_WIDTH, a.FRAME_HEIGHT))
    assert np.array_equal(data.map1, m1) and np.array_equal(data.map2, m2)
    imgs = W.synthetic_frames(1, a.FRAME_WIDTH, a.FRAME_HEIGHT, seed=21, kind="random").reshape(4, a.FRAME_HEIGHT, a.FRAME_WIDTH, 3)
    got = cal.undistort_batch(imgs)
    for i in range(4):
        assert np.array_equal(got[i], oracle.remap(imgs[i], m1, m2))
    assert np.array_equal(cal.undistort(imgs[0]), got[0])
    a.FRAME_WIDTH, a.FRAME_HEIGHT, a.FOCAL_SCALE, a.SIZE_SCALE = 1280, 1024, 0.5, 1


def test_incalibrator_undistort_width_not_a_multiple_of_4(ffi, oracle):
    """cv2.remap through the tile plan with a destination pitch that is not a multiple of 4 pixels (padded rows + compaction)."""
    from cameracalibration_amd.IntrinsicCalibration import InCalibrator

    a = InCalibrator.get_args()
    keep = (a.FRAME_WIDTH, a.FRAME_HEIGHT, a.FOCAL_SCALE, a.SIZE_SCALE)
    try:
        for fw, fh in [(322, 250), (318, 252)]:
            a.FRAME_WIDTH, a.FRAME_HEIGHT, a.FOCAL_SCALE, a.SIZE_SCALE = fw, fh, 0.6, 1
            K, D = W.undistort_calibration()
            K = np.diag([fw / 1280.0, fh / 960.0, 1.0]) @ K
            cal = InCalibrator("fisheye")
            data = cal.set_calibration(K, D)
            imgs = np.random.default_rng(8).integers(0, 256, (5, fh, fw, 3), dtype=np.uint8)
            got = cal.undistort_batch(imgs)
            for i in range(5):
                assert np.array_equal(got[i], oracle.remap(imgs[i], data.map1, data.map2)), (fw, fh, i)
    finally:
        a.FRAME_WIDTH, a.FRAME_HEIGHT, a.FOCAL_SCALE, a.SIZE_SCALE = keep


def test_incalibrator_on_reference_image(ffi, oracle, repo_rig):
    from cameracalibration_amd.IntrinsicCalibration import InCalibrator

    img = repo_rig.image("incalib")
    K, D, _ = repo_rig.rig["front"]
    cal = InCalibrator("fisheye")
    cal.set_calibration(K, D)
    Kd = oracle.camera_mat_dst(K, 1280, 1024, 0.5, 1)
    m1, m2 = oracle.fisheye_init_undistort_rectify_map(K, D, Kd, (1280, 1024))
    assert np.array_equal(cal.undistort(img), oracle.remap(img, m1, m2))


def test_excalibrator_warp(ffi, oracle, repo_rig):
    from cameracalibration_amd.ExtrinsicCalibration import ExCalibrator

    src = repo_rig.image("excalib_src")  # 2560x2048
    H = repo_rig.rig["back"][2]
    ex = ExCalibrator()
    ex.set_homography(H, src, (1000, 1000))
    assert np.array_equal(ex.warp(), oracle.warp_perspective(src, H, (1000, 1000)))
    rng = np.random.default_rng(4)
    small = rng.integers(0, 256, (40, 90, 3), dtype=np.uint8)
    for Hm, size in [(np.eye(3), (90, 40)), (np.array([[1, 0, 5.0], [0, 1, 3.0], [0, 0, 1]]), (90, 40)),
                     (np.array([[0.9, 0.2, -3.0], [-0.1, 1.1, 2.0], [1e-3, -2e-3, 1.0]]), (131, 67)),
                     (np.array([[1, 2, 3.0], [2, 4, 6.0], [1, 1, 1.0]]), (20, 10))]:  # last one is singular
        ex.set_homography(Hm, small, (size[1], size[0]))
        assert np.array_equal(ex.warp(), oracle.warp_perspective(small, Hm, size))


def test_center_image_translate(ffi, oracle, repo_rig):
    """CenterImage.translate (extrinsicCalib.py:54-59)"""
    from cameracalibration_amd.ExtrinsicCalibration.extrinsicCalib import CenterImage

    img = repo_rig.image("front")
    h, w = img.shape[:2]
    for (x, y) in [(w // 2, h // 2), (100, 900), (w - 1, 0), (0, 7), (3 * w, -h)]:
        c = CenterImage().set_center(x, y)
        assert np.array_equal(c.translate(img), oracle.translate(img, w // 2 - x, h // 2 - y)), (x, y)
    assert CenterImage()(img) is img                      # (0, 0): the reference returns the frame untouched
    assert np.array_equal(CenterImage().set_center(5, 6)(img), oracle.translate(img, w // 2 - 5, h // 2 - 6))


@pytest.mark.parametrize("f", [0.25, 0.37, 0.5, 0.93, 1.0, 1.6, 2.0, 3.3])
def test_resize_linear_bit_exact(ffi, oracle, f):
    """cv2.resize(img, (0,0), fx=f, fy=f) (ScaleImage.__call__, extrinsicCalib.py:125) through the C-ABI."""
    rng = np.random.default_rng(int(f * 100))
    for (h, w) in [(61, 83), (1, 7), (9, 2), (240, 320)]:
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        ds = np.zeros(2, np.int32)
        st = ffi.lib().bevw_resize_dsize(w, h, f, f, ffi.ptr(ds))
        want_shape = (int(np.rint(h * f)), int(np.rint(w * f)))
        if min(want_shape) <= 0:
            assert st == -1
            continue
        assert st == 0 and (int(ds[1]), int(ds[0])) == want_shape
        out = np.empty((int(ds[1]), int(ds[0]), 3), np.uint8)
        ffi.check(ffi.lib().bevw_resize_linear_u8c3(0, ffi.ptr(img), w, h, f, f, 1, ffi.ptr(out)))
        assert np.array_equal(out, oracle.resize_linear(img, f, f)), (h, w, f)
    # anisotropic factors and a batch of two
    img = rng.integers(0, 256, (2, 50, 70, 3), dtype=np.uint8)
    ds = np.zeros(2, np.int32)
    ffi.check(ffi.lib().bevw_resize_dsize(70, 50, f, 1.0 / f, ffi.ptr(ds)))
    out = np.empty((2, int(ds[1]), int(ds[0]), 3), np.uint8)
    ffi.check(ffi.lib().bevw_resize_linear_u8c3(0, ffi.ptr(img), 70, 50, f, 1.0 / f, 2, ffi.ptr(out)))
    for b in range(2):
        assert np.array_equal(out[b], oracle.resize_linear(img[b], f, 1.0 / f))


@pytest.mark.parametrize("square", [4.0, 10.0, 23.5])
def test_scale_image_mirror(ffi, oracle, repo_rig, square):
    """ScaleImage (extrinsicCalib.py:87-130): board-square distance -> scale factor -> resize -> pad / centre-crop."""
    from cameracalibration_amd.ExtrinsicCalibration import extrinsicCalib as EC

    a = EC.ExCalibrator.get_args()
    bw, bh = a.BORAD_WIDTH, a.BORAD_HEIGHT
    gx, gy = np.meshgrid(np.arange(bw), np.arange(bh))
    corners = np.stack([100 + gx * square, 50 + gy * square], -1).reshape(-1, 1, 2).astype(np.float32)
    sc = EC.ScaleImage(corners)
    assert abs(sc.dist_square - square) < 1e-9 and sc.scale_factor == a.SCALED_SIZE / sc.dist_square
    img = repo_rig.image("back")[:301, :403]
    h, w = img.shape[:2]
    got = sc(img)
    ref = oracle.resize_linear(img, sc.scale_factor, sc.scale_factor)
    if sc.scale_factor < 1:
        want = np.zeros_like(img)
        t, l = (h - ref.shape[0]) // 2, (w - ref.shape[1]) // 2
        want[t:t + ref.shape[0], l:l + ref.shape[1]] = ref
    else:
        t, l = (ref.shape[0] - h) // 2, (ref.shape[1] - w) // 2
        want = ref[t:t + h, l:l + w]
    assert got.shape == img.shape and np.array_equal(got, want)


@pytest.mark.parametrize("seed", range(48))
def test_random_rigs_modes_and_batches(ffi, SB, oracle, seed):
    """Seeded fuzz over the geometry: frame / BEV / car sizes (odd ones included, so every schedule and every alignment
    fallback is hit), focal and size scales, modes, batch sizes, with and without the car -- always bit-exact."""
    rng = np.random.default_rng(1000 + seed)
    fw = int(rng.choice([160, 200, 236, 320, 322, 400]))
    fh = int(rng.choice([128, 150, 256, 258]))
    bw = int(rng.choice([96, 124, 125, 200, 248, 250]))
    bh = int(rng.choice([96, 130, 201, 250]))
    cw, ch = int(rng.integers(0, bw // 3 + 1)), int(rng.integers(0, bh // 2 + 1))
    cfg = dict(FRAME_WIDTH=fw, FRAME_HEIGHT=fh, BEV_WIDTH=bw, BEV_HEIGHT=bh, CAR_WIDTH=cw, CAR_HEIGHT=ch,
               FOCAL_SCALE=float(rng.choice([0.8, 1.0, 1.25])), SIZE_SCALE=float(rng.choice([1.0, 1.5, 2.0])))
    # the repo rig scaled to the frame (raw frame by fw/1280, fh/1024) and to the BEV (by bw/1000, bh/1000)
    A = np.diag([fw / 1280.0, fh / 1024.0, 1.0])
    U = np.diag([fw * cfg["SIZE_SCALE"] / 2560.0, fh * cfg["SIZE_SCALE"] / 2048.0, 1.0])   # undistorted grid scale
    Bm = np.diag([bw / 1000.0, bh / 1000.0, 1.0])
    rig = {n: (A @ K, D.copy(), Bm @ H @ np.linalg.inv(U)) for n, (K, D, H) in W.repo_rig().items()}
    blend, balance = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    batch = int(rng.integers(1, 6))
    frames = rng.integers(0, 256, (batch, 4, fh, fw, 3), dtype=np.uint8)
    frames[:, 1] //= 2                       # unequal brightness: non-trivial luminance deltas
    car = None
    if rng.integers(0, 2) and cw and ch:
        car = np.zeros((bh, bw, 3), np.uint8)
        t, l = (bh - ch) // 2, (bw - cw) // 2
        car[t:t + ch, l:l + cw] = rng.integers(0, 256, (ch, cw, 3), dtype=np.uint8)
    bev, ref = make_pair(SB, oracle, rig, cfg, blend, balance, schedule=int(rng.choice([0, 0, 1])))
    got = bev.batch(frames, car)
    for b in range(batch):
        want = ref(*frames[b], car=car)
        assert np.array_equal(got[b], want), "seed %d cfg %s blend %d balance %d frame %d: %d bytes differ" % (
            seed, cfg, blend, balance, b, np.count_nonzero(got[b] != want))


# ---------------------------------------------------------------------------------------------------------------
# full BASELINE sizes: size-independent properties + spot parity
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("blend,balance", [(False, False), (True, True)])
def test_full_size_batch_properties(ffi, SB, oracle, blend, balance):
    cfg, rig = W.CONFIG_S, W.rig_s()
    batch, uniq = 64, 2
    frames = W.synthetic_frames(uniq, cfg["FRAME_WIDTH"], cfg["FRAME_HEIGHT"], seed=W.SEED)
    bev, ref = make_pair(SB, oracle, rig, cfg, blend, balance)
    assert bev.plan_info()["schedule"] == 2
    d_in = ffi.DeviceBuffer(batch * frames[0].nbytes)
    pitch = bev.out_pitch   # the default device layout: rows of whole 64-byte sectors (1088 pixels) on the tile plan
    assert pitch == 1088
    d_out = ffi.DeviceBuffer(batch * 1080 * pitch * 3)
    d_out.fill(0xA5)
    for b in range(batch):
        d_in.upload(frames[b % uniq], offset=b * frames[0].nbytes)
    bev.run_device(d_in.ptr, batch, None, d_out.ptr, out_bytes=d_out.nbytes)
    bev.sync()
    out = np.ascontiguousarray(d_out.download((batch, 1080, pitch, 3))[:, :, :1080])
    # a frame's BEV does not depend on its position in the batch (XCD / chunk mapping, batch loop)
    for b in range(uniq, batch):
        assert np.array_equal(out[b], out[b % uniq]), b
    # and equals the oracle
    for b in range(uniq):
        assert maxdiff(out[b], ref(*frames[b])) == 0
    # the per-pixel schedule gives the same bytes
    bev1 = SB.BevGenerator(blend=blend, balance=balance, rig=rig, schedule=1)
    assert np.array_equal(bev1.batch(frames), out[:uniq])
    # ragged batch sizes through the host-buffer entry point
    rag = bev.batch(np.concatenate([frames, frames, frames[:1]]))
    assert np.array_equal(rag[:2], out[:2]) and np.array_equal(rag[4], out[0])


@pytest.mark.parametrize("blend", [False, True])
def test_bench_configuration_batch_256_all_random(ffi, SB, oracle, blend):
    """The bench's own configuration (config S, batch 256 frame sets resident in HBM, what BENCH_rNN.json times) on the
    WORST-CASE input for parity: every byte of every frame uniform random, 3 distinct frame sets cycled through the batch.
    Every one of the 256 BEVs must equal the oracle's BEV of its frame set -- bit-exact."""
    cfg, rig = W.CONFIG_S, W.rig_s()
    batch, uniq = 256, 3
    frames = W.synthetic_frames(uniq, cfg["FRAME_WIDTH"], cfg["FRAME_HEIGHT"], seed=W.SEED + 17, kind="random")
    bev, ref = make_pair(SB, oracle, rig, cfg, blend, False)
    assert bev.plan_info()["schedule"] == 2
    d_in = ffi.DeviceBuffer(batch * frames[0].nbytes)
    pitch = bev.out_pitch   # the bench's layout = the product's default: rows of 1088 pixels on the device
    d_out = ffi.DeviceBuffer(batch * 1080 * pitch * 3)
    d_out.fill(0x5A)
    for b in range(batch):
        d_in.upload(frames[b % uniq], offset=b * frames[0].nbytes)
    bev.run_device(d_in.ptr, batch, None, d_out.ptr, out_bytes=d_out.nbytes)
    bev.sync()
    want = [ref(*frames[u]) for u in range(uniq)]
    per = 1080 * pitch * 3
    for b0 in range(0, batch, 32):   # download in slices: the whole batch is 0.9 GB
        out = d_out.download((32, 1080, pitch, 3), offset=b0 * per)
        for k in range(32):
            assert np.array_equal(out[k][:, :1080], want[(b0 + k) % uniq]), "frame %d of the batch differs from the oracle" % (b0 + k)
    d_in.free()
    d_out.free()


def test_bench_configuration_4_blend_balance_batch_256_all_random(ffi, SB, oracle):
    """BASELINE config 4 exactly as bench.py --workload blend_balance_b256 times it: BevGenerator(blend=True, balance=True), 256 frame
    sets resident in HBM, run_device (the engine cuts the batch into 2 slices of 128 over its two streams), the default device layout.
    Worst-case input: every byte uniform random, 3 distinct frame sets cycled through the batch.  Every one of the 256 BEVs must equal
    the oracle's luminance_balance -> warp -> blend -> cv2.add -> color_balance result of its frame set -- bit-exact."""
    cfg, rig = W.CONFIG_S, W.rig_s()
    batch, uniq = 256, 3
    frames = W.synthetic_frames(uniq, cfg["FRAME_WIDTH"], cfg["FRAME_HEIGHT"], seed=W.SEED + 29, kind="random")
    bev, ref = make_pair(SB, oracle, rig, cfg, True, True)
    assert bev.plan_info()["schedule"] == 2
    d_in = ffi.DeviceBuffer(batch * frames[0].nbytes)
    pitch = bev.out_pitch
    d_out = ffi.DeviceBuffer(batch * bev.out_image_bytes)
    d_out.fill(0x5A)
    for b in range(batch):
        d_in.upload(frames[b % uniq], offset=b * frames[0].nbytes)
    for _ in range(2):   # twice: the second step re-uses every scratch buffer of the first (pre-gain BEV, per-unit sums, shifted groups)
        bev.run_device(d_in.ptr, batch, None, d_out.ptr, out_bytes=d_out.nbytes)
    bev.sync()
    want = [ref(*frames[u]) for u in range(uniq)]
    per = bev.out_image_bytes
    for b0 in range(0, batch, 32):
        out = d_out.download((32, 1080, pitch, 3), offset=b0 * per)
        for k in range(32):
            assert np.array_equal(out[k][:, :1080], want[(b0 + k) % uniq]), "frame %d of the batch differs from the oracle" % (b0 + k)
    d_in.free()
    d_out.free()


def test_bench_configuration_2_undistort_batch_64_all_random(ffi, oracle):
    """BASELINE config 2 exactly as bench.py --workload undistort_b64 times it: bevw_fisheye_remapper_create + bevw_remap_device on 64
    images resident in HBM.  Every image uniform random and DISTINCT; every output compared with oracle.remap through the oracle's own
    fisheye maps (cv2.fisheye.initUndistortRectifyMap + cv2.remap, intrinsicCalib.py:193-195) -- bit-exact."""
    import ctypes as C
    ucfg = W.CONFIG_UNDISTORT
    K, D = W.undistort_calibration()
    fw, fh = ucfg["FRAME_WIDTH"], ucfg["FRAME_HEIGHT"]
    L = ffi.lib()
    r = C.c_void_p()
    ffi.check(L.bevw_fisheye_remapper_create(0, fw, fh, ffi.ptr(ffi.f64(K, 9)), ffi.ptr(ffi.f64(D, 4)), ucfg["FOCAL_SCALE"], ucfg["SIZE_SCALE"],
                                             0.0, 0.0, C.byref(r)))
    try:
        batch = 64
        imgs = np.random.default_rng(W.SEED + 31).integers(0, 256, (batch, fh, fw, 3), dtype=np.uint8)
        Kd = oracle.camera_mat_dst(K, fw, fh, ucfg["FOCAL_SCALE"], ucfg["SIZE_SCALE"])
        m1, m2 = oracle.fisheye_init_undistort_rectify_map(K, D, Kd, (int(fw * ucfg["SIZE_SCALE"]), int(fh * ucfg["SIZE_SCALE"])))
        dh, dw = m1.shape[:2]
        d_in = ffi.DeviceBuffer(imgs.nbytes).upload(imgs)
        d_out = ffi.DeviceBuffer(batch * dh * dw * 3)
        d_out.fill(0x5A)
        ffi.check(L.bevw_remap_device(r, d_in.ptr, batch, d_out.ptr))
        ffi.check(L.bevw_remapper_sync(r))
        out = d_out.download((batch, dh, dw, 3))
        for b in range(batch):
            assert np.array_equal(out[b], oracle.remap(imgs[b], m1, m2)), "image %d of the batch differs from the oracle" % b
        d_in.free()
        d_out.free()
    finally:
        L.bevw_remapper_destroy(r)


def test_4k_rig_spot(ffi, SB, oracle):
    cfg, rig = W.CONFIG_4K, W.rig_4k()
    frames = W.synthetic_frames(1, cfg["FRAME_WIDTH"], cfg["FRAME_HEIGHT"], seed=9, kind="random")
    bev, ref = make_pair(SB, oracle, rig, cfg, True, False)
    assert maxdiff(bev.batch(frames)[0], ref(*frames[0])) == 0


def test_main_py_runbev_body_drop_in(ffi, oracle, repo_rig, tmp_path):
    """main.py:72-89 (runBEV) with the package on sys.path the way the reference lays it out, K/D/H read from .npy files."""
    import subprocess
    import sys

    from conftest import ROOT
    import os

    for n in CAMS:
        os.makedirs(tmp_path / "data" / n)
        for kind, arr in zip("KDH", repo_rig.rig[n]):
            np.save(tmp_path / "data" / n / f"camera_{n}_{kind}.npy", arr)
        np.save(tmp_path / f"{n}.npy", repo_rig.image(n))
    code = f"""
import sys, numpy as np
sys.path.insert(0, {os.path.join(ROOT, 'cameracalibration_amd')!r})
from SurroundBirdEyeView import BevGenerator
front, back, left, right = (np.load({str(tmp_path)!r} + '/%s.npy' % n) for n in ('front', 'back', 'left', 'right'))
args = BevGenerator.get_args()
args.CAR_WIDTH = 200
args.CAR_HEIGHT = 350
bev = BevGenerator(blend=True, balance=True)
surround = bev(front, back, left, right)
np.save({str(tmp_path)!r} + '/surround.npy', surround)
"""
    env = dict(os.environ, BEVW_DATA_DIR=str(tmp_path / "data"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=str(tmp_path))
    assert out.returncode == 0, out.stderr
    cfg = dict(W.CONFIG_R, CAR_WIDTH=200, CAR_HEIGHT=350)
    want = oracle.RefBevGenerator(repo_rig.rig, cfg, blend=True, balance=True)(*repo_rig.frames())
    assert np.array_equal(np.load(tmp_path / "surround.npy"), want)


@pytest.mark.parametrize("blend", [False, True])
def test_mask_call_standalone(ffi, SB, oracle, blend):
    """Mask.__call__ / BlendMask.__call__ (surroundBEV.py:161-162, 279-280) outside the fused stitch."""
    rig = small_rig()
    bev, ref = make_pair(SB, oracle, rig, SMALL_CFG, blend, False)
    rng = np.random.default_rng(8)
    img = rng.integers(0, 256, (SMALL_CFG["BEV_HEIGHT"], SMALL_CFG["BEV_WIDTH"], 3), dtype=np.uint8)
    for i in range(4):
        assert np.array_equal(bev.masks[i](img), ref.apply_mask(i, img)), i


def test_incalibrator_normal_pinhole(ffi, oracle, repo_rig):
    """InCalibrator('normal') (intrinsicCalib.py:150-163, 193-195): pinhole undistort maps + remap, 5 and 8 coefficients."""
    from cameracalibration_amd.IntrinsicCalibration import InCalibrator

    img = repo_rig.image("incalib")  # 1280 x 1024
    K = np.array([[820.0, 0, 655.3], [0, 815.5, 500.7], [0, 0, 1]])
    for D in ([-0.31, 0.12, 0.0011, -0.0007, -0.02], [-0.2, 0.05, 0.001, 0.002, 0.01, 0.02, -0.01, 0.003], [0.05, -0.01, 0, 0]):
        cal = InCalibrator("normal")
        data = cal.set_calibration(K, D)
        Kd = oracle.camera_mat_dst(K, 1280, 1024, 0.5, 1)
        m1, m2 = oracle.init_undistort_rectify_map(K, D, Kd, (1280, 1024))
        assert np.array_equal(data.map1, m1) and np.array_equal(data.map2, m2)
        assert np.array_equal(cal.undistort(img), oracle.remap(img, m1, m2))


# ---------------------------------------------------------------------------------------------------------------
# device-side row pitch (bevw_set_output_pitch): rows of whole 64-byte sectors; host results stay dense
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("blend,balance", [(False, False), (True, False), (False, True), (True, True)])
def test_output_pitch_aligned_config_s(ffi, SB, oracle, blend, balance):
    """BevGenerator(output_pitch='aligned') on the bench rig (1080-pixel rows -> 1088): the host entry points return the same dense
    arrays as ever (rows are compacted inside the device-to-host copy), run_device writes [B][BH][1088][3] whose first 1080 columns
    equal the oracle bit for bit -- all four modes, with a car sprite, batch 5 (a ragged chunk)."""
    cfg, rig = W.CONFIG_S, W.rig_s()
    set_args(SB, cfg)
    bw, bh = cfg["BEV_WIDTH"], cfg["BEV_HEIGHT"]
    frames = W.synthetic_frames(2, cfg["FRAME_WIDTH"], cfg["FRAME_HEIGHT"], seed=W.SEED + 5, kind="random")
    rng = np.random.default_rng(11)
    car = np.zeros((bh, bw, 3), np.uint8)
    car[300:780, 380:700] = rng.integers(0, 256, (480, 320, 3), dtype=np.uint8)
    bev = SB.BevGenerator(blend=blend, balance=balance, rig=rig, output_pitch='aligned')
    ref = oracle.RefBevGenerator(rig, cfg, blend=blend, balance=balance)
    assert bev.out_pitch == 1088 and bev.plan_info()["schedule"] == 2
    want = [ref(*frames[u], car) for u in range(2)]
    assert np.array_equal(bev(*frames[0], car), want[0])                       # reference call shape, dense result
    batch = 5
    host = bev.batch(np.stack([frames[b % 2] for b in range(batch)]), car)     # host batch, dense result
    for b in range(batch):
        assert np.array_equal(host[b], want[b % 2]), "host batch frame %d" % b
    d_in = ffi.DeviceBuffer(batch * frames[0].nbytes)
    d_car = ffi.DeviceBuffer(car.nbytes).upload(car)
    d_out = ffi.DeviceBuffer(batch * bh * bev.out_pitch * 3)
    d_out.fill(0x5A)
    for b in range(batch):
        d_in.upload(frames[b % 2], offset=b * frames[0].nbytes)
    bev.run_device(d_in.ptr, batch, d_car.ptr, d_out.ptr, out_bytes=d_out.nbytes)
    bev.sync()
    out = d_out.download((batch, bh, bev.out_pitch, 3))
    for b in range(batch):
        assert np.array_equal(out[b, :, :bw], want[b % 2]), "device frame %d" % b
    for buf in (d_in, d_car, d_out):
        buf.free()


@pytest.mark.parametrize("parts,ring", [(8, 1), (5, 1), (7, 0)])
def test_balance_slices_with_ring_buffers(ffi, parts, ring):
    """BEVW_BAL_PARTS / BEVW_BAL_RING (switches read once per process, hence the child): config 4's chain in many small slices over the handle's
    two streams, the compact scratch and the pre-gain BEV as per-stream ring buffers that every second slice overwrites (round 6's
    Infinity-Cache A/B).  Uneven slices (40 frame sets in 7 or 8 parts), a sprite, the device-resident entry: same bytes as the oracle."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    code = (
        "import numpy as np, sys\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "from cameracalibration_amd import workloads as W\n"
        "from cameracalibration_amd.SurroundBirdEyeView import surroundBEV as SB\n"
        "from oracle import oracle as O\n"
        "import test_gpu_parity as T\n"
        "rig = T.small_rig(); T.set_args(SB, T.SMALL_CFG)\n"
        "frames = W.synthetic_frames(40, T.SMALL_CFG['FRAME_WIDTH'], T.SMALL_CFG['FRAME_HEIGHT'], kind='random')\n"
        "car = np.random.default_rng(5).integers(0, 200, (T.SMALL_CFG['BEV_HEIGHT'], T.SMALL_CFG['BEV_WIDTH'], 3), dtype=np.uint8)\n"
        "bev = SB.BevGenerator(blend=True, balance=True, rig=rig)\n"
        "ref = O.RefBevGenerator(rig, T.SMALL_CFG, blend=True, balance=True)\n"
        "for rnd in range(2):\n"
        "    got = bev.batch(frames, car if rnd else None)\n"
        "    assert bev.plan_info()['schedule'] == 2\n"
        "    assert all(np.array_equal(got[b], ref(*frames[b], car if rnd else None)) for b in range(40)), rnd\n"
        "print('balance slices ok')\n") % (ROOT, os.path.join(ROOT, "tests"))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, BEVW_BAL_PARTS=str(parts), BEVW_BAL_RING=str(ring)), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "balance slices ok" in r.stdout, r.stdout + r.stderr


def test_device_copy_yardstick(ffi):
    """bevw_device_copy_rate: what a plain copy kernel moves (read + write) on this box -- the measured figure bench.py prints beside the 8 TB/s
    specification peak.  Sanity only: above 1 TB/s, below the specification."""
    for streaming in (False, True):
        g = ffi.device_copy_rate(256 << 20, 5, streaming)
        assert 1000.0 < g < 8000.0, (streaming, g)
    assert ffi.lib().bevw_device_copy_rate(0, 8, 1, 0, None) != 0


def test_run_device_on_a_pitched_handle_wants_the_buffer_size(ffi, SB, oracle):
    """the default device layout has rows of whole sectors (1088 pixels for a 1080-pixel BEV); the library sees raw pointers, so a
    caller that sized its buffer for dense images must fail loudly instead of being overrun: run_device refuses a pitched handle without
    out_bytes and any buffer that is too small; dense handles keep the reference-shaped call."""
    cfg, rig = SMALL_CFG, small_rig()
    set_args(SB, cfg)
    bw, bh = cfg["BEV_WIDTH"], cfg["BEV_HEIGHT"]
    frames = W.synthetic_frames(1, cfg["FRAME_WIDTH"], cfg["FRAME_HEIGHT"], seed=5, kind="random")
    d_in = ffi.DeviceBuffer(frames.nbytes).upload(frames)
    bev = SB.BevGenerator(rig=rig)                               # 'auto' -> aligned: 248 -> 256 pixels per row
    assert bev.out_pitch == 256 and bev.out_image_bytes == 256 * bh * 3
    d_dense = ffi.DeviceBuffer(bw * bh * 3)
    with pytest.raises(Exception, match="out_bytes"):
        bev.run_device(d_in.ptr, 1, None, d_dense.ptr)
    with pytest.raises(Exception, match="need"):
        bev.run_device(d_in.ptr, 1, None, d_dense.ptr, out_bytes=d_dense.nbytes)
    d_ok = ffi.DeviceBuffer(bev.out_image_bytes)
    bev.run_device(d_in.ptr, 1, None, d_ok.ptr, out_bytes=d_ok.nbytes)
    bev.sync()
    want = oracle.RefBevGenerator(rig, cfg, blend=False, balance=False)(*frames[0])
    assert np.array_equal(d_ok.download((bh, 256, 3))[:, :bw], want)
    dense = SB.BevGenerator(rig=rig, output_pitch='dense')
    dense.run_device(d_in.ptr, 1, None, d_dense.ptr)            # the reference-shaped call: no size needed
    dense.sync()
    assert np.array_equal(d_dense.download((bh, bw, 3)), want)
    for b in (d_in, d_dense, d_ok):
        b.free()


def test_output_pitch_small_rig_and_errors(ffi, SB, oracle):
    """An explicit pitch on the small rig (248 -> 272 pixels), the dense default, and the refusals: a pitch below the width, not a
    multiple of 4, or together with the per-pixel schedule."""
    cfg, rig = SMALL_CFG, small_rig()
    set_args(SB, cfg)
    frames = W.synthetic_frames(1, cfg["FRAME_WIDTH"], cfg["FRAME_HEIGHT"], seed=3, kind="random")
    ref = oracle.RefBevGenerator(rig, cfg, blend=True, balance=True)
    want = ref(*frames[0])
    for pitch in ('dense', 'aligned', 272):
        bev = SB.BevGenerator(blend=True, balance=True, rig=rig, output_pitch=pitch)
        assert bev.out_pitch == {'dense': 248, 'aligned': 256, 272: 272}[pitch]
        assert np.array_equal(bev(*frames[0]), want), pitch
    for bad in (244, 250):
        with pytest.raises(Exception):
            SB.BevGenerator(rig=rig, output_pitch=bad)
    with pytest.raises(Exception):
        SB.BevGenerator(rig=rig, output_pitch='aligned', schedule=ffi.SCHED_PER_PIXEL)
