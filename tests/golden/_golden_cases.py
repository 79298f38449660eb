"""Case list shared by make_goldens_with_cv2.py (computes each case with the real cv2 + the reference's own modules) and
tests/test_cv2_goldens.py (computes the same case with the oracle).  A case is a name and a uint8 / int16 / uint16 array."""
import hashlib

import numpy as np

CAMS = ("front", "back", "left", "right")
MODES = [(False, False), (True, False), (False, True), (True, True)]
RESIZE_FACTORS = (0.37, 0.5, 1.6, 2.0)
MAIN_CAR = (200, 350)          # main.py:80-81
PINHOLE_D = (-0.31, 0.12, 0.0015, -0.0007, -0.02)
# cv2.addWeighted(channel, gain, 0, 0, 0) as color_balance calls it (surroundBEV.py:52-54: src2 is the Python scalar 0, which is what
# selects the work type inside OpenCV).  Three gains on which a CV_32F and a CV_64F evaluation differ for at least one byte value,
# applied to the ramp 0..255: the case decides the `addWeighted` switch by itself, independent of any image content.
ADDWEIGHTED_GAINS = (0.9241071147452891, 1.3978102017483192, 0.7298850131855613)


def digest(arr: np.ndarray) -> str:
    a = np.ascontiguousarray(arr)
    return hashlib.sha256(str(a.dtype).encode() + str(a.shape).encode() + a.tobytes()).hexdigest()


def probe(arr: np.ndarray, n: int = 4096) -> np.ndarray:
    """Deterministic strided sample for diagnostics when a digest differs."""
    flat = np.ascontiguousarray(arr).reshape(-1)
    step = max(1, flat.size // n)
    return flat[::step][:n].copy()


def pack(cases: dict) -> dict:
    blob = {}
    for name, arr in cases.items():
        blob[name + "__sha"] = np.array(digest(arr))
        blob[name + "__shape"] = np.array(arr.shape, np.int64)
        blob[name + "__probe"] = probe(arr)
    return blob


def jpeg_test_image() -> np.ndarray:
    """A deterministic 1000 x 1000 BGR image (the reference's BEV size: 62.5 MCUs per side, dummy blocks on both edges) for the cv2.imwrite cases."""
    y, x = np.mgrid[0:1000, 0:1000]
    a = np.stack([(x * 3 + y) % 256, 128 + 100 * np.sin(x / 37.0) * np.cos(y / 23.0), (x * y // 7) % 256], -1)
    return np.ascontiguousarray(np.clip(a, 0, 255).astype(np.uint8))
