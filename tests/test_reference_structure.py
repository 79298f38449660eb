"""Runs the REFERENCE's own Python modules (from /root/reference) with `cv2` replaced by the oracle's primitives
(tests/_cv2_shim.py) and compares with the oracle's restatement of the whole path: same frames in, same bytes out.
This pins the structure of the restatement -- operation order, mask polygons, seam lines, blend-weight loop, padding,
luminance / white balance composition, pad / crop rules -- against the reference itself.  The reference tree exists only
in the build container: everywhere else these tests skip (nothing here runs on the GPU box)."""
import importlib.util
import os
import sys

import numpy as np
import pytest

REF = os.environ.get("BEVW_REFERENCE_ROOT", "/root/reference")
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present")

CAMS = ("front", "back", "left", "right")


def load_reference_module(relpath, name):
    """Import one reference file as a module with the oracle-built cv2 and an empty argv (they parse args at import)."""
    import _cv2_shim

    _cv2_shim.install()
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, relpath))
    mod = importlib.util.module_from_spec(spec)
    argv, sys.argv = sys.argv, [sys.argv[0]]
    try:
        spec.loader.exec_module(mod)
    finally:
        sys.argv = argv
    return mod


@pytest.fixture(scope="module")
def ref_sb():
    mod = load_reference_module("SurroundBirdEyeView/surroundBEV.py", "reference_surroundBEV")
    yield mod
    sys.modules.pop("cv2", None)


def test_reference_rig_files_equal_the_fixture(ref_sb, repo_rig):
    cam = ref_sb.Camera.__new__(ref_sb.Camera)
    for n in CAMS:
        base = os.path.join(REF, "SurroundBirdEyeView", "data", n)
        for kind, want in zip("KDH", repo_rig.rig[n]):
            assert np.array_equal(np.load(os.path.join(base, "camera_%s_%s.npy" % (n, kind))), want)
    del cam


@pytest.mark.parametrize("blend,balance,car_size", [(False, False, (250, 400)), (True, True, (200, 350)), (True, False, (250, 400)),
                                                    (False, True, (200, 350))])
def test_reference_control_flow_equals_the_restatement(ref_sb, oracle, repo_rig, blend, balance, car_size):
    args = ref_sb.BevGenerator.get_args()
    args.CAR_WIDTH, args.CAR_HEIGHT = car_size            # main.py:79-81 uses 200 x 350, the module default is 250 x 400
    gen = ref_sb.BevGenerator(blend=blend, balance=balance)
    cfg = dict(oracle.DEFAULT_CFG, CAR_WIDTH=car_size[0], CAR_HEIGHT=car_size[1])
    ref = oracle.RefBevGenerator(repo_rig.rig, cfg, blend=blend, balance=balance)
    for i, n in enumerate(CAMS):
        assert np.array_equal(gen.cameras[i].camera_mat_dst, ref.cameras[i].K_dst), n
        assert np.array_equal(gen.cameras[i].bev_maps[0], ref.cameras[i].bev_maps[0]), n
        assert np.array_equal(gen.cameras[i].bev_maps[1], ref.cameras[i].bev_maps[1]), n
        assert np.array_equal(gen.masks[i].mask, ref.masks[i]), n + " mask"
        if blend:
            assert np.array_equal(gen.masks[i].weight, np.repeat(ref.weights[i][:, :, None], 3, axis=2)
                                  if ref.weights[i].ndim == 2 else ref.weights[i]), n + " weight"
    frames = repo_rig.frames()
    car = ref_sb.padding(repo_rig.image("car"), cfg["BEV_WIDTH"], cfg["BEV_HEIGHT"])
    assert np.array_equal(car, oracle.padding(repo_rig.image("car"), cfg["BEV_WIDTH"], cfg["BEV_HEIGHT"]))
    for c in (None, car):
        got = gen(*[f.copy() for f in frames], car=c)
        want = ref(*frames, car=c)
        assert got.dtype == np.uint8 and got.shape == want.shape
        assert np.array_equal(got, want), "%d bytes differ" % np.count_nonzero(got != want)


def test_reference_balance_helpers_equal_the_restatement(ref_sb, oracle, repo_rig):
    frames = repo_rig.frames()
    got = ref_sb.luminance_balance([f.copy() for f in frames])
    want = oracle.luminance_balance(frames)
    for g, w in zip(got, want):
        assert np.array_equal(g, w)
    img = repo_rig.image("back")
    assert np.array_equal(ref_sb.color_balance(img.copy()), oracle.color_balance(img))


def test_reference_excalib_preprocessing_equals_the_mirror_rules(oracle, repo_rig):
    ec = load_reference_module("ExtrinsicCalibration/extrinsicCalib.py", "reference_extrinsicCalib")
    try:
        img = repo_rig.image("back")[:301, :403]
        h, w = img.shape[:2]
        center = ec.CenterImage.__new__(ec.CenterImage)
        center.x, center.y = 100, 250
        assert np.array_equal(center.translate(img), oracle.translate(img, w // 2 - 100, h // 2 - 250))
        a = ec.args
        gx, gy = np.meshgrid(np.arange(a.BORAD_WIDTH), np.arange(a.BORAD_HEIGHT))
        for square in (4.0, 23.5):
            corners = np.stack([100 + gx * square, 50 + gy * square], -1).reshape(-1, 2).astype(np.float32)
            sc = ec.ScaleImage(corners)
            ref = oracle.resize_linear(img, sc.scale_factor, sc.scale_factor)
            if sc.scale_factor < 1:
                want = np.zeros_like(img)
                t, l = (h - ref.shape[0]) // 2, (w - ref.shape[1]) // 2
                want[t:t + ref.shape[0], l:l + ref.shape[1]] = ref
            else:
                t, l = (ref.shape[0] - h) // 2, (ref.shape[1] - w) // 2
                want = ref[t:t + h, l:l + w]
            assert np.array_equal(sc(img), want), square
    finally:
        sys.modules.pop("cv2", None)
