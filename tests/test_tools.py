"""Tools/ drivers around the hot path."""
import os

import numpy as np
import pytest

from cameracalibration_amd.Tools import timeAlign as TA


def test_align_time_all_frames_present():
    base = [10.0 + 0.5 * i for i in range(6)]
    stamps = {"front": [t + 0.01 for t in base], "back": [t - 0.02 for t in base], "left": [t + 0.03 for t in base],
              "right": [t + 0.04 for t in base]}
    groups, cams = TA.align_time(stamps, 0.1)
    assert cams[0] == "right" and sorted(cams) == ["back", "front", "left", "right"]   # latest first stamp seeds
    assert all(len(g) == 4 for g in groups) and len(groups) == 6
    for g, t in zip(groups, base):
        assert max(g) - min(g) < 0.1 and abs(sum(g) / 4 - t) < 0.05


def test_align_time_dropped_and_extra_frames():
    seed = [1.0, 2.0, 3.0, 4.0]
    stamps = {"a": seed, "b": [0.2, 0.99, 3.02, 3.5, 4.01], "c": [0.98, 2.01, 2.5, 4.03, 9.0]}
    groups, cams = TA.align_time(stamps, 0.1)
    assert cams == ["a", "b", "c"]
    assert groups[0] == [1.0, 0.99, 0.98]          # 0.2 is older than every group: dropped
    assert groups[1] == [2.0, 2.01]                # b has no frame near 2.0
    assert groups[2] == [3.0, 3.02]                # c's 2.5 matches nothing and is skipped
    assert groups[3] == [4.0, 4.01, 4.03]          # 3.5 skipped, 9.0 runs past the last group
    # extending an existing alignment (init=False)
    groups2, cams2 = TA.align_time({"d": [1.05, 3.95]}, 0.1, init=False, info_list=[groups, cams])
    assert cams2[-1] == "d" and groups2[0][-1] == 1.05 and groups2[3][-1] == 3.95


def test_time_parser_reads_directories(tmp_path):
    import argparse

    ns = argparse.Namespace(usb_align_thresh=0.1)
    for cam, off in (("front", 0.0), ("back", 0.01), ("left", 0.02), ("right", 0.03)):
        d = tmp_path / cam
        d.mkdir()
        for t in (100.0, 100.5, 101.0):
            if cam == "left" and t == 100.5:
                continue                              # one camera misses a frame: that group is incomplete
            (d / f"{t + off:.3f}.jpg").write_bytes(b"")
        setattr(ns, cam, str(d))
    res, cams = TA.TimeParser(ns).usb_cam_align()
    assert len(res) == 2 and all(len(g) == 4 for g in res) and cams[0] == "right"


@pytest.mark.gpu
def test_undistort_tool_matches_oracle(oracle, repo_rig, tmp_path):
    """Tools/undistort.py:25-77 end to end: K/D from .npy, optical-axis offsets, size scale, PNG round trip."""
    from PIL import Image

    from cameracalibration_amd.Tools import undistort as U

    K, D, _ = repo_rig.rig["front"]
    np.save(tmp_path / "K.npy", K)
    np.save(tmp_path / "D.npy", D)
    src, dst = tmp_path / "in", tmp_path / "out"
    src.mkdir(); dst.mkdir()
    img = repo_rig.image("front")
    for i in range(3):
        Image.fromarray(np.ascontiguousarray(np.roll(img, 7 * i, axis=1)[:, :, ::-1])).save(src / f"f{i}.png")
    n = U.main(["-path_read", str(src) + "/", "-path_save", str(dst) + "/", "-path_k", str(tmp_path / "K.npy"),
                "-path_d", str(tmp_path / "D.npy"), "-srcformat", "png", "-dstformat", "png", "-focalscale", "0.8",
                "-sizescale", "1.5", "-offset_h", "12.5", "-offset_v", "-8", "-quality", "3"])
    assert n == 3
    Kd = oracle.camera_mat_dst(K, 1280, 1024, 0.8, 1.5, 12.5, -8.0)
    m1, m2 = oracle.fisheye_init_undistort_rectify_map(K, D, Kd, (int(1280 * 1.5), int(1024 * 1.5)))
    for i in range(3):
        got = np.asarray(Image.open(dst / f"f{i}.png").convert("RGB"))[:, :, ::-1]
        assert np.array_equal(got, oracle.remap(np.ascontiguousarray(np.roll(img, 7 * i, axis=1)), m1, m2)), i
    und = U.Undistorter(U.DEFAULT_K, U.DEFAULT_D, 1280, 1024)
    assert np.array_equal(und.maps()[0], oracle.fisheye_init_undistort_rectify_map(
        U.DEFAULT_K, U.DEFAULT_D, oracle.camera_mat_dst(U.DEFAULT_K, 1280, 1024, 1, 1), (1280, 1024))[0])


@pytest.mark.gpu
def test_undistort_tool_with_jpeg_files_in_and_out(oracle, repo_rig, tmp_path):
    """Tools/undistort.py:60-77 with its default formats: cv2.imread of .jpg files, cv2.imwrite(.jpg, quality).  The GPU codec reads and
    writes them; the files must be the ones libjpeg-turbo (Pillow) writes for the oracle's pixels.  A progressive file in the directory is
    outside the codec's subset and goes through Pillow; a file of another size is skipped."""
    import io

    from PIL import Image

    from cameracalibration_amd.Tools import undistort as U
    from tests import _jpeg_common as JC

    K, D, _ = repo_rig.rig["front"]
    np.save(tmp_path / "K.npy", K)
    np.save(tmp_path / "D.npy", D)
    src, dst = tmp_path / "in", tmp_path / "out"
    src.mkdir(); dst.mkdir()
    cams = JC.repo_camera_jpegs()
    for n in ("front", "left"):
        (src / f"{n}.jpg").write_bytes(cams[n])                       # the reference's own files
    b = io.BytesIO()
    Image.fromarray(np.ascontiguousarray(repo_rig.image("back")[:, :, ::-1])).save(b, "JPEG", quality=90, progressive=True)
    (src / "prog.jpg").write_bytes(b.getvalue())
    (src / "small.jpg").write_bytes(JC.pil_encode(JC.image(64, 80, 2)))
    n = U.main(["-path_read", str(src) + "/", "-path_save", str(dst) + "/", "-path_k", str(tmp_path / "K.npy"),
                "-path_d", str(tmp_path / "D.npy"), "-quality", "92"])
    assert n == 3
    Kd = oracle.camera_mat_dst(K, 1280, 1024, 1, 1)
    m1, m2 = oracle.fisheye_init_undistort_rectify_map(K, D, Kd, (1280, 1024))
    for name, raw in (("front", cams["front"]), ("left", cams["left"]), ("prog", b.getvalue())):
        want = JC.pil_encode(oracle.remap(JC.pil_decode(raw), m1, m2), 92)
        assert (dst / f"{name}.jpg").read_bytes() == want, name
    assert not (dst / "small.jpg").exists()


def test_align_time_matches_reference_goldens():
    """tests/golden/time_align.json was produced by the reference's own align_time (make_time_align_goldens.py)."""
    import json

    from conftest import ROOT

    cases = json.load(open(os.path.join(ROOT, "tests", "golden", "time_align.json")))
    assert len(cases) == 60
    for c in cases:
        td = {k: list(c["time_dict"][k]) for k in c["order_in"]}   # dict order matters: it is the merge order
        groups, cams = TA.align_time(td, c["thresh"])
        assert cams == c["cams"]
        assert groups == c["groups"]
