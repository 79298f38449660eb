#!/usr/bin/env python3
"""Writes tests/golden/jpeg_goldens.npz: small JPEG files made by libjpeg-turbo (Pillow's build -- the library behind cv2.imread / cv2.imwrite,
main.py:74-77, surroundBEV.py:340) with the SHA-256 of what the SAME library decodes them to, and small images with the SHA-256 of the file the
library writes for them.  tests/test_jpeg_goldens.py holds the oracle (CPU) and the HIP codec (GPU) against these without importing Pillow.

    python tests/golden/make_jpeg_goldens.py      (needs Pillow; rerun only to extend the cases)
"""
import hashlib
import io
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import PIL  # noqa: E402
from PIL import Image, features  # noqa: E402

from tests import _jpeg_common as JC  # noqa: E402

DECODE_CASES = [   # name, (h, w), kind, quality, Pillow subsampling, extra save arguments
    ("d420_camera_like", (120, 168), 2, 90, 2, {}),
    ("d420_noise_q100", (48, 80), 1, 100, 2, {}),
    ("d420_odd", (37, 53), 2, 75, 2, {}),
    ("d422", (64, 97), 0, 85, 1, {}),
    ("d444", (33, 48), 2, 60, 0, {}),
    ("d420_restart3", (96, 160), 2, 88, 2, {"restart_marker_blocks": 3}),
    ("d420_restart_rows", (96, 160), 1, 70, 2, {"restart_marker_rows": 1}),
    ("d420_private_tables", (136, 200), 2, 85, 2, {"optimize": True}),
    ("d420_tiny", (1, 1), 2, 95, 2, {}),
    ("d420_flat", (64, 64), 3, 95, 2, {}),
]
ENCODE_CASES = [   # name, (h, w), kind, quality, sampling byte, Pillow subsampling
    ("e420_q95", (72, 104), 2, 95, 0x22, 2),
    ("e420_odd_q50", (37, 53), 0, 50, 0x22, 2),
    ("e422_q80", (40, 56), 2, 80, 0x21, 1),
    ("e444_q100", (17, 33), 1, 100, 0x11, 0),
    ("e420_q10", (100, 75), 2, 10, 0x22, 2),
]


def image(h, w, kind):
    if kind == 3:
        return np.full((h, w, 3), (200, 30, 120), np.uint8)
    return JC.image(h, w, kind)


def main():
    out = {"made_with": np.array("Pillow %s, libjpeg-turbo %s" % (PIL.__version__, features.version("jpg")))}
    for name, (h, w), kind, q, sub, kw in DECODE_CASES:
        f = JC.pil_encode(image(h, w, kind), q, sub, **kw)
        out["dec_" + name] = np.frombuffer(f, np.uint8)
        out["dec_" + name + "_sha"] = np.array(hashlib.sha256(JC.pil_decode(f).tobytes()).hexdigest())
    b = io.BytesIO()
    Image.fromarray(image(50, 70, 2)[:, :, 0]).save(b, "JPEG", quality=90)
    out["dec_grey"] = np.frombuffer(b.getvalue(), np.uint8)
    out["dec_grey_sha"] = np.array(hashlib.sha256(JC.pil_decode(b.getvalue()).tobytes()).hexdigest())
    for name, (h, w), kind, q, samp, sub in ENCODE_CASES:
        im = image(h, w, kind)
        out["enc_" + name] = im
        out["enc_" + name + "_params"] = np.array([q, samp], np.int32)
        out["enc_" + name + "_sha"] = np.array(hashlib.sha256(JC.pil_encode(im, q, sub)).hexdigest())
    path = os.path.join(ROOT, "tests", "golden", "jpeg_goldens.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(DECODE_CASES) + 1, "decode +", len(ENCODE_CASES), "encode cases")


if __name__ == "__main__":
    main()
