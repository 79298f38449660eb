"""Row f4, the oracle's pin: oracle/jpegoracle.c against Pillow's libjpeg-turbo -- the library behind cv2.imread / cv2.imwrite
(main.py:74-77, surroundBEV.py:340) -- byte for byte.  CPU only."""
from __future__ import annotations

import glob
import os

import numpy as np
import pytest

from tests import _jpeg_common as JC

pytest.importorskip("PIL")
from oracle import jpeg as JO  # noqa: E402


def test_decode_equals_libjpeg_turbo_on_the_reference_camera_files():
    for name, raw in JC.repo_camera_jpegs().items():
        info = JO.probe(raw)
        assert (info["width"], info["height"], info["components"], info["h_samp"], info["v_samp"]) == (1280, 1024, 3, 2, 2), name
        assert np.array_equal(JO.imdecode(raw), JC.pil_decode(raw)), name


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="reference tree absent")
def test_decode_equals_libjpeg_turbo_on_every_jpeg_of_the_reference_tree():
    n = 0
    for p in sorted(glob.glob("/root/reference/**/*.jpg", recursive=True)):
        raw = open(p, "rb").read()
        if raw[:2] != b"\xff\xd8":
            continue   # SurroundBirdEyeView/data/car.jpg is a PNG
        assert np.array_equal(JO.imdecode(raw), JC.pil_decode(raw)), p
        n += 1
    assert n >= 16


@pytest.mark.parametrize("sub,samp", JC.SUBSAMPLINGS)
def test_decode_and_encode_equal_libjpeg_turbo(sub, samp):
    for h, w in JC.SIZES:
        for kind in (0, 1, 2):
            for q in (95, 50, 100, 10):
                im = JC.image(h, w, kind)
                f = JC.pil_encode(im, q, sub)
                assert np.array_equal(JO.imdecode(f), JC.pil_decode(f)), (h, w, kind, q)
                assert JO.imencode(im, q, samp) == f, (h, w, kind, q)


def test_the_bev_size_file_is_the_file_libjpeg_turbo_writes():
    im = JC.image(1080, 1080, 2)   # surroundBEV.py:340 writes a BEV_HEIGHT x BEV_WIDTH image; 1080 = 67.5 MCUs: dummy blocks on both edges
    f = JC.pil_encode(im)
    assert JO.imencode(im) == f
    assert np.array_equal(JO.imdecode(f), JC.pil_decode(f))


def test_grayscale_and_restart_intervals():
    im = JC.image(50, 70, 2)
    f = JC.pil_encode_gray(im[:, :, 0])
    assert JO.probe(f)["components"] == 1
    assert np.array_equal(JO.imdecode(f), JC.pil_decode(f))
    for kw, ri in ((dict(restart_marker_blocks=1), 1), (dict(restart_marker_blocks=3), 3), (dict(restart_marker_rows=1), 5)):
        f = JC.pil_encode(im, 90, 2, **kw)
        assert JO.probe(f)["restart_interval"] == ri
        assert np.array_equal(JO.imdecode(f), JC.pil_decode(f)), kw


def test_seeded_random_cases_against_libjpeg_turbo():
    """300 random (size, content, quality, sampling, restart interval, private tables) cases: decode == Pillow's decode, encode == Pillow's file."""
    import io

    from PIL import Image

    rng = np.random.default_rng(2024)
    done = 0
    for case in range(300):
        h, w = int(rng.integers(1, 260)), int(rng.integers(1, 260))
        sub, samp = JC.SUBSAMPLINGS[int(rng.integers(0, 3))]
        q = int(rng.choice([3, 25, 50, 75, 90, 95, 100]))
        kind = int(rng.integers(0, 4))
        im = np.full((h, w, 3), rng.integers(0, 256, 3), np.uint8) if kind == 3 else JC.image(h, w, kind, seed=case)
        kw = {}
        if rng.random() < 0.3:
            kw["restart_marker_blocks"] = int(rng.integers(1, 30))
        if rng.random() < 0.3:
            kw["optimize"] = True
        b = io.BytesIO()
        try:
            Image.fromarray(np.ascontiguousarray(im[:, :, ::-1])).save(b, "JPEG", quality=q, subsampling=sub, **kw)
        except OSError:
            continue   # Pillow's own output buffer is too small for some tiny images with extra markers
        f = b.getvalue()
        assert np.array_equal(JO.imdecode(f), JC.pil_decode(f)), (case, h, w, q, sub, kw)
        if not kw:
            assert JO.imencode(im, q, samp) == f, (case, h, w, q, sub)
        done += 1
    assert done > 250


def test_out_of_scope_files_are_refused():
    from PIL import Image
    import io

    im = JC.image(32, 32, 2)
    b = io.BytesIO()
    Image.fromarray(im).save(b, "JPEG", progressive=True)
    with pytest.raises(ValueError):
        JO.probe(b.getvalue())
    with pytest.raises(ValueError):
        JO.probe(b"\x89PNG\r\n\x1a\n" + bytes(64))
    with pytest.raises(ValueError):
        JO.probe(JC.pil_encode(im)[:300])   # truncated inside the tables
