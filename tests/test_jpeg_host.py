"""Row f4, host logic without a GPU: the marker parser behind bevw_jpeg_probe (csrc/bevw_jpeg.h parse_header) -- what it accepts, what it
refuses and why -- through the C-ABI; the compute entry points fail loudly when no device is visible."""
import io
import os

import numpy as np
import pytest

from tests import _jpeg_common as JC

pytest.importorskip("PIL")


@pytest.fixture(scope="module")
def IC():
    from cameracalibration_amd import build, imgcodecs

    build.build()
    return imgcodecs


def test_probe_reads_the_reference_camera_files(IC):
    for name, raw in JC.repo_camera_jpegs().items():
        info = IC.probe(raw)
        assert (info["width"], info["height"], info["components"], info["h_samp"], info["v_samp"], info["restart_interval"]) == (1280, 1024, 3, 2, 2, 0), name
        assert info["orientation"] in (0, 1)


def test_probe_fields_follow_the_file(IC):
    im = JC.image(50, 70, 2)
    for sub, (h, v) in ((2, (2, 2)), (1, (2, 1)), (0, (1, 1))):
        info = IC.probe(JC.pil_encode(im, 80, sub, restart_marker_blocks=3))
        assert (info["width"], info["height"], info["h_samp"], info["v_samp"], info["restart_interval"]) == (70, 50, h, v, 3)
    g = IC.probe(JC.pil_encode_gray(im[:, :, 0]))
    assert (g["components"], g["h_samp"], g["v_samp"]) == (1, 1, 1)


def test_probe_refuses_by_name(IC):
    from PIL import Image
    from cameracalibration_amd._ffi import BevwError

    im = JC.image(48, 64, 2)
    rgb = Image.fromarray(np.ascontiguousarray(im[:, :, ::-1]))

    def save(**kw):
        b = io.BytesIO()
        rgb.save(b, "JPEG", **kw)
        return b.getvalue()
    with pytest.raises(BevwError, match="progressive"):
        IC.probe(save(progressive=True))
    b = io.BytesIO()
    rgb.convert("CMYK").save(b, "JPEG")
    with pytest.raises(BevwError, match="component count"):
        IC.probe(b.getvalue())
    exif = Image.Exif()
    exif[0x0112] = 6     # turned by 90 degrees: the decoder applies the tag as cv2.imread does, and the probe reports the size of the RESULT
    info = IC.probe(save(exif=exif))
    assert (info["orientation"], info["width"], info["height"], info["stored_width"], info["stored_height"]) == (6, 48, 64, 64, 48)
    exif[0x0112] = 3
    info = IC.probe(save(exif=exif))
    assert (info["orientation"], info["width"], info["height"]) == (3, 64, 48)
    exif[0x0112] = 1
    assert IC.probe(save(exif=exif))["orientation"] == 1
    exif[0x0112] = 9     # not a valid orientation: ignored
    assert IC.probe(save(exif=exif))["orientation"] == 1
    good = save()
    with pytest.raises(BevwError, match="not a JPEG"):
        IC.probe(b"\x89PNG\r\n\x1a\n" + bytes(64))
    with pytest.raises(BevwError, match="truncated"):
        IC.probe(good[:200])
    with pytest.raises(BevwError):
        IC.probe(good[: good.index(b"\xff\xda") + 3])     # cut inside the scan header


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="a GPU is present: the no-device path is not reachable")
def test_codec_fails_loudly_without_a_device(IC):
    from cameracalibration_amd._ffi import BevwError

    with pytest.raises(BevwError, match="no HIP device|CPU"):
        IC.JpegCodec(0)
    with pytest.raises(BevwError):
        IC.imdecode(JC.pil_encode(JC.image(16, 16, 2)))
