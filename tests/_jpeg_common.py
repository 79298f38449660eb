"""Shared helpers of the JPEG tests (row f4): synthetic images, Pillow as the libjpeg-turbo reference, the repo's camera JPEGs."""
from __future__ import annotations

import io
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def image(h: int, w: int, kind: int, seed: int = 0) -> np.ndarray:
    """BGR uint8 test image: 0 = integer pattern, 1 = uniform noise (worst case for the entropy coder), 2 = smooth + noise (camera-like)."""
    rng = np.random.default_rng(seed * 7919 + h * 131 + w * 17 + kind)
    y, x = np.mgrid[0:h, 0:w]
    if kind == 0:
        a = np.stack([(x * 3 + y) % 256, (x + y * 2) % 256, (x * y) % 256], -1)
    elif kind == 1:
        a = rng.integers(0, 256, (h, w, 3))
    else:
        a = (128 + 100 * np.sin(x / 7.0)[..., None] * np.cos(y / 5.0)[..., None] * np.array([1, 0.5, -1]) + rng.normal(0, 8, (h, w, 3))).clip(0, 255)
    return np.ascontiguousarray(a.astype(np.uint8))


def pil_decode(raw: bytes) -> np.ndarray:
    """What cv2.imread gives for this file: libjpeg-turbo's default decode (ISLOW IDCT, fancy upsampling), as BGR."""
    from PIL import Image

    return np.ascontiguousarray(np.asarray(Image.open(io.BytesIO(raw)).convert("RGB"))[:, :, ::-1])


def pil_encode(bgr: np.ndarray, quality: int = 95, subsampling: int = 2, **kw) -> bytes:
    """The file libjpeg-turbo writes with cv2.imwrite's settings (quality 95, 4:2:0 = Pillow subsampling 2, no optimisation)."""
    from PIL import Image

    b = io.BytesIO()
    Image.fromarray(np.ascontiguousarray(bgr[:, :, ::-1])).save(b, "JPEG", quality=quality, subsampling=subsampling, **kw)
    return b.getvalue()


def pil_encode_gray(gray: np.ndarray, quality: int = 90, **kw) -> bytes:
    from PIL import Image

    b = io.BytesIO()
    Image.fromarray(np.ascontiguousarray(gray)).save(b, "JPEG", quality=quality, **kw)
    return b.getvalue()


def repo_camera_jpegs() -> dict:
    """The reference's four camera files (SurroundBirdEyeView/data/*/*.jpg, 1280 x 1024, baseline 4:2:0) from the committed fixture."""
    z = np.load(os.path.join(ROOT, "tests", "golden", "repo_rig.npz"))
    return {n: z[f"{n}_img"].tobytes() for n in ("front", "back", "left", "right")}


SUBSAMPLINGS = ((2, 0x22), (1, 0x21), (0, 0x11))   # Pillow's code, the SOF sampling byte of the luma component
SIZES = ((8, 8), (16, 16), (1, 1), (3, 5), (17, 33), (40, 56), (100, 75), (64, 2), (2, 64), (5, 4), (4, 5), (33, 48))
