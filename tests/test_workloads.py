"""Host logic: workload definitions, the reference-API mirror's argument handling.  No GPU."""
import os

import numpy as np
import pytest

from cameracalibration_amd import workloads as W


def test_rig_constants_match_the_fixture(repo_rig):
    rig = W.repo_rig()
    for n in W.CAMERA_NAMES:
        for got, want in zip(rig[n], repo_rig.rig[n]):
            assert np.array_equal(got.reshape(-1), np.asarray(want, np.float64).reshape(-1)), n


def test_derived_rigs():
    r, s, k4 = W.repo_rig(), W.rig_s(), W.rig_4k()
    for n in W.CAMERA_NAMES:
        assert s[n][0][1, 2] == r[n][0][1, 2] - 32 and s[n][0][0, 0] == r[n][0][0, 0]
        # H_S maps (x, y) of the cropped undistort grid like H maps (x, y + 64), scaled by 1.08
        p = np.array([700.0, 900.0, 1.0])
        a = s[n][2] @ p
        b = r[n][2] @ (p + [0, 64, 0])
        assert np.allclose(a[:2] / a[2], 1.08 * b[:2] / b[2])
        assert k4[n][0][0, 0] == 3 * r[n][0][0, 0] and k4[n][0][1, 2] == 3 * r[n][0][1, 2] - 456
        q = np.array([2100.0, 2000.0, 1.0])
        a = k4[n][2] @ q
        b = r[n][2] @ np.array([700.0, 2000.0 / 3 + 304, 1.0])
        assert np.allclose(a[:2] / a[2], 1.08 * b[:2] / b[2])


def test_synthetic_frames_are_seeded_and_shaped():
    a = W.synthetic_frames(1, 64, 48)
    b = W.synthetic_frames(1, 64, 48)
    assert a.shape == (1, 4, 48, 64, 3) and a.dtype == np.uint8 and np.array_equal(a, b)
    assert not np.array_equal(a, W.synthetic_frames(1, 64, 48, seed=W.SEED + 1))
    c = W.synthetic_frames(2, 16, 8, kind="constant")
    assert (c == c[:, :, :1, :1]).all()
    assert W.synthetic_frames(1, 16, 8, kind="random").std() > 60


def test_algorithmic_bytes_table():
    assert W.ALGORITHMIC_BYTES["direct_stitch_b256"] == 2_033_157 + 1080 * 1080 * 3
    assert W.ALGORITHMIC_BYTES["undistort_b64"] == 1_735_512 + 1280 * 960 * 3
    assert W.ALGORITHMIC_BYTES["blend_balance_b256"] == 4 * 1280 * 960 * 3 + 2 * 2_170_338 + 1080 * 1080 * 3


def test_mirror_namespace_matches_reference_defaults():
    from cameracalibration_amd.SurroundBirdEyeView import surroundBEV as SB
    from cameracalibration_amd.SurroundBirdEyeView import BevGenerator

    ns = SB.parser.parse_args([])
    assert vars(ns) == dict(FRAME_WIDTH=1280, FRAME_HEIGHT=1024, BEV_WIDTH=1000, BEV_HEIGHT=1000, CAR_WIDTH=250,
                            CAR_HEIGHT=400, FOCAL_SCALE=1, SIZE_SCALE=2, BLEND_FLAG=False, BALANCE_FLAG=False)
    assert BevGenerator.get_args() is SB.args
    with pytest.raises(Exception, match="name should be front/back/left/right"):
        SB.Camera("top", np.eye(3), np.zeros(4), np.eye(3))
    with pytest.raises(Exception, match="name should be front/back/left/right"):
        SB.Mask("top")


def test_padding_matches_oracle(oracle):
    from cameracalibration_amd.SurroundBirdEyeView.surroundBEV import padding

    rng = np.random.default_rng(0)
    for (h, w, H, Wd) in [(400, 250, 1000, 1000), (7, 5, 10, 12), (3, 4, 8, 9)]:
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        assert np.array_equal(padding(img, Wd, H), oracle.padding(img, Wd, H))


def test_calibrator_mirrors_argument_errors():
    from cameracalibration_amd.IntrinsicCalibration import InCalibrator
    from cameracalibration_amd.ExtrinsicCalibration import ExCalibrator

    with pytest.raises(Exception, match="camera should be fisheye/normal"):
        InCalibrator("pinhole")
    cal = InCalibrator("fisheye")
    a = InCalibrator.get_args()
    assert (a.FRAME_WIDTH, a.FRAME_HEIGHT, a.FOCAL_SCALE, a.SIZE_SCALE) == (1280, 1024, 0.5, 1)
    with pytest.raises(Exception, match="no calibration"):
        cal.undistort(np.zeros((1024, 1280, 3), np.uint8))
    with pytest.raises(Exception, match="no homography"):
        ExCalibrator().warp()


def test_bench_shard_sizes():
    import bench

    assert bench.shard_sizes(256, 8) == [32] * 8
    assert bench.shard_sizes(10, 4) == [3, 3, 2, 2] and sum(bench.shard_sizes(257, 8)) == 257


def test_drop_in_top_level_import(tmp_path):
    """main.py:5-7 imports `SurroundBirdEyeView`, `IntrinsicCalibration`, `ExtrinsicCalibration` as top-level packages."""
    import subprocess
    import sys

    from conftest import ROOT

    code = ("import sys; sys.path.insert(0, %r); "
            "from SurroundBirdEyeView import BevGenerator; from IntrinsicCalibration import InCalibrator; "
            "from ExtrinsicCalibration import ExCalibrator; a = BevGenerator.get_args(); a.CAR_WIDTH = 200; "
            "print(a.CAR_WIDTH, InCalibrator.get_args().FOCAL_SCALE, ExCalibrator.get_args().CAMERA_ID)"
            % os.path.join(ROOT, "cameracalibration_amd"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=str(tmp_path))
    assert out.returncode == 0, out.stderr
    assert out.stdout.split() == ["200", "0.5", "1"]
