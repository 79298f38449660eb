"""A stand-in for the `cv2` module built from the ORACLE's primitives -- TEST INFRASTRUCTURE ONLY.

Purpose: run the REFERENCE's own Python control flow (imported from /root/reference, where that tree exists) on top of the
oracle's restatement of each OpenCV primitive, and compare the result with the oracle's restatement of the whole
generator.  That pins the STRUCTURE of the restatement (operation order, polygons, seam lines, padding, the balance
composition) against the reference itself; the arithmetic of the primitives stays pinned by the known-answer tests
(and by tests/golden/make_goldens_with_cv2.py wherever a real cv2 exists).  Only what the reference's hot-path modules
call is provided; anything else raises AttributeError.
"""
import sys
import types

import numpy as np

from oracle import oracle as O

IS_ORACLE_SHIM = True
CV_16SC2, INTER_LINEAR, BORDER_CONSTANT = 11, 1, 0
COLOR_BGR2HSV, COLOR_HSV2BGR = 40, 54
WINDOW_NORMAL, WINDOW_KEEPRATIO, NORM_L2 = 0, 0, 4
EVENT_LBUTTONDOWN, EVENT_MOUSEMOVE, EVENT_LBUTTONUP, EVENT_FLAG_LBUTTON = 1, 0, 4, 1
FONT_HERSHEY_PLAIN = 1
TERM_CRITERIA_EPS, TERM_CRITERIA_MAX_ITER = 2, 1
CALIB_CB_ADAPTIVE_THRESH, CALIB_CB_FAST_CHECK, CALIB_CB_NORMALIZE_IMAGE = 1, 8, 2
CALIB_FIX_K3 = 128
IMREAD_COLOR, IMWRITE_JPEG_QUALITY = 1, 1


class _Any(types.SimpleNamespace):
    """cv2.fisheye.CALIB_* flags are read at import time by intrinsicCalib.py; any integer will do."""

    def __getattr__(self, name):
        if name.startswith("CALIB_"):
            return 0
        raise AttributeError(name)


def _fisheye_maps(K, D, R, P, size, m1type):
    assert m1type == CV_16SC2 and np.array_equal(np.asarray(R, float), np.eye(3))
    return O.fisheye_init_undistort_rectify_map(K, np.asarray(D, float).reshape(-1), P, tuple(size))


fisheye = _Any(initUndistortRectifyMap=_fisheye_maps)


def initUndistortRectifyMap(K, D, R, P, size, m1type):
    assert m1type == CV_16SC2 and np.array_equal(np.asarray(R, float), np.eye(3))
    return O.init_undistort_rectify_map(K, np.asarray(D, float).reshape(-1), P, tuple(size))


def remap(src, map1, map2, interpolation=INTER_LINEAR):
    assert interpolation == INTER_LINEAR
    return O.remap(src, map1, map2)


def warpPerspective(src, M, dsize):
    return O.warp_perspective(src, M, tuple(dsize))


def warpAffine(src, M, dsize):
    M = np.asarray(M, np.float64)
    assert np.array_equal(M[:, :2], np.eye(2)) and np.array_equal(M[:, 2], np.rint(M[:, 2])), "integer shifts only"
    assert tuple(dsize) == (src.shape[1], src.shape[0])
    return O.translate(src, int(M[0, 2]), int(M[1, 2]))


def resize(src, dsize, fx=0.0, fy=0.0):
    assert tuple(dsize) == (0, 0)
    return O.resize_linear(src, fx, fy)


def fillPoly(img, pts, color):
    assert len(pts) == 1
    return O.fill_poly(img, np.asarray(pts[0], np.int32), int(color))


def bitwise_and(a, b, mask=None):
    if mask is None:
        return np.bitwise_and(a, b)
    assert a is b or np.array_equal(a, b)
    out = np.empty_like(a)
    O.lib().orc_mask_select(O._p(np.ascontiguousarray(a)), O._p(np.ascontiguousarray(mask)), a.size // 3, O._p(out))
    return out


def pointPolygonTest(contour, pt, measureDist):
    assert measureDist
    line = np.ascontiguousarray(np.asarray(contour, np.int32).reshape(4))
    return O.lib().orc_segment_distance(O._p(line), float(pt[0]), float(pt[1]))


def split(img):
    return [np.ascontiguousarray(img[..., c]) for c in range(img.shape[2])]


def merge(planes):
    return np.ascontiguousarray(np.stack(planes, axis=-1))


def cvtColor(img, code):
    return {COLOR_BGR2HSV: O.bgr2hsv, COLOR_HSV2BGR: O.hsv2bgr}[code](img)


def add(a, b):
    if isinstance(b, np.ndarray):
        return O.add_sat(a, b)
    # 8U array + Python scalar: the scalar is rounded once, then a saturating integer add
    d = O.lib().orc_round_delta(float(b))
    return np.clip(a.astype(np.int32) + d, 0, 255).astype(np.uint8)


def addWeighted(src1, alpha, src2, beta, gamma, dst=None):
    assert src2 == 0 and beta == 0 and gamma == 0, "only the reference's gain form"
    tmp = np.ascontiguousarray(np.repeat(src1.reshape(-1, 1), 3, axis=1))
    g = np.array([alpha, alpha, alpha], np.float64)
    O.lib().orc_gain(O._p(tmp), tmp.shape[0], O._p(g))
    out = tmp[:, 0].reshape(src1.shape)
    if dst is not None:
        dst[...] = out
        return dst
    return out


def copyMakeBorder(img, top, bottom, left, right, borderType, value=(0, 0, 0)):
    assert borderType == BORDER_CONSTANT and tuple(value)[:3] == (0, 0, 0)
    out = np.zeros((img.shape[0] + top + bottom, img.shape[1] + left + right) + img.shape[2:], img.dtype)
    out[top:top + img.shape[0], left:left + img.shape[1]] = img
    return out


def norm(a, b, normType=NORM_L2):
    assert normType == NORM_L2
    return float(np.sqrt(((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2).sum()))


def imdecode(buf, flags=IMREAD_COLOR):
    from oracle import jpeg as JO

    assert flags == IMREAD_COLOR
    return JO.imdecode(np.asarray(buf, np.uint8).tobytes())


def imencode(ext, img, params=None):
    from oracle import jpeg as JO

    assert ext == ".jpg"
    q = 95
    if params:
        for k, v in zip(params[0::2], params[1::2]):
            if k == IMWRITE_JPEG_QUALITY:
                q = int(v)
    return True, np.frombuffer(JO.imencode(np.ascontiguousarray(img), q), np.uint8)


def install():
    """Make `import cv2` resolve to this module (only when no real cv2 is importable)."""
    sys.modules["cv2"] = sys.modules[__name__]
    return sys.modules[__name__]
