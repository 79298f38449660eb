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


# ---- implementation probes: classic fixed-point / float-weight paths against the kernels OpenCV >= 4.11 introduced -------------------------
# OpenCV 4.11 rewrote warpPerspective (new SIMD kernels for 8U / 16U / 32F, C1 / C3 / C4) and reworked remap in the same release line.  The
# reference goes through both: cv2.warpPerspective of the 16-bit undistort maps (surroundBEV.py:105-108 -> the BEV look-up table, the "LUT quirk"),
# of an 8UC3 image (extrinsicCalib.py:166-169), cv2.remap of 8UC3 frames (surroundBEV.py:113-117).  These cases are SMALL and stored WHOLE
# (pack_full), on inputs chosen so that the two families of kernels give different bytes: a hash only says "differs", the array says which
# implementation produced it (tests/test_cv2_goldens.py::test_implementation_probes names it).
PROBE_H = ((0.91, -0.13, 7.3), (0.08, 1.07, -4.1), (2.3e-4, -1.7e-4, 1.0))   # a homography with a perspective row: every pixel its own fraction
PROBE_SIZE = (96, 80)       # source (w, h)
PROBE_DSIZE = (120, 90)     # destination (w, h)


def probe_images():
    """uint16 (one channel, the map2 shape: values 0 .. 1023), int16 (two channels, the map1 shape) and uint8 BGR noise + gradients."""
    rng = np.random.default_rng(411)
    w, h = PROBE_SIZE
    y, x = np.mgrid[0:h, 0:w]
    u16 = ((x * 37 + y * 91) % 1024).astype(np.uint16)
    u16[::3, ::5] = rng.integers(0, 1024, u16[::3, ::5].shape, dtype=np.uint16)
    s16 = np.stack([x * 27 - 900, y * 31 - 1200], -1).astype(np.int16)
    s16[1::4, 2::3] += rng.integers(-50, 50, s16[1::4, 2::3].shape, dtype=np.int16)
    u8 = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    u8[h // 2:] = np.stack([(x * 5) % 256, (y * 7) % 256, (x + y) % 256], -1).astype(np.uint8)[h // 2:]
    return u16, s16, u8


def probe_maps():
    """16SC2 + 16UC1 maps of a smooth warp with every one of the 1024 fraction codes present: a cv2.remap probe independent of warpPerspective."""
    w, h = PROBE_DSIZE
    y, x = np.mgrid[0:h, 0:w]
    fx = (x * 0.77 + y * 0.11 + 1.5) * 32.0
    fy = (y * 0.83 - x * 0.05 + 2.25) * 32.0
    ix, iy = np.rint(fx).astype(np.int64), np.rint(fy).astype(np.int64)
    m1 = np.stack([ix >> 5, iy >> 5], -1).astype(np.int16)
    m2 = ((iy & 31) * 32 + (ix & 31)).astype(np.uint16)
    return m1, m2


def pack_full(cases: dict) -> dict:
    return {name + "__full": np.ascontiguousarray(arr) for name, arr in cases.items()}

