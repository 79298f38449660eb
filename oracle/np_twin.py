"""Second, independent (vectorised NumPy) statement of the oracle's numeric primitives.  TEST INFRASTRUCTURE ONLY.

Purpose: catch transcription slips in oracle/bevoracle.c.  Both follow SURVEY.md Appendix A; neither is pinned to a
real cv2 (PARITY UNPINNED).  Written array-at-a-time so the code shape shares nothing with the C loops.
"""
from __future__ import annotations

import numpy as np

Q = 32


def _rne(a):
    return np.rint(a)  # round-half-to-even, like cvRound


def fisheye_map(K, D, Knew, size):
    """A.1: fp64 projection per undistorted pixel, Q5 split (surroundBEV.py:98-103)."""
    w, h = size
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    k = np.asarray(D, np.float64).reshape(-1)
    ifx, ify = 1.0 / Knew[0, 0], 1.0 / Knew[1, 1]
    x0, y0 = -Knew[0, 2] / Knew[0, 0], -Knew[1, 2] / Knew[1, 1]
    rows = np.arange(h, dtype=np.float64)
    # column walk is an accumulation (+= iR00), reproduced with a sequential ufunc.accumulate
    steps = np.full((h, w), ifx, np.float64)
    steps[:, 0] = rows * 0.0 + x0
    xs = np.add.accumulate(steps, axis=1)
    ys = np.repeat((rows * ify + y0)[:, None], w, axis=1)
    ws = np.repeat((rows * 0.0 + 1.0)[:, None], w, axis=1)
    x, y = xs / ws, ys / ws
    r = np.sqrt(x * x + y * y)
    th = np.arctan(r)
    t2 = th * th
    t4 = t2 * t2
    t6 = t4 * t2
    t8 = t4 * t4
    thd = th * (1 + k[0] * t2 + k[1] * t4 + k[2] * t6 + k[3] * t8)
    with np.errstate(invalid="ignore", divide="ignore"):
        scale = np.where(r == 0, 1.0, thd / r)
    u = fx * x * scale + cx
    v = fy * y * scale + cy
    iu = _rne(u * Q).astype(np.int64)
    iv = _rne(v * Q).astype(np.int64)
    m1 = np.stack([(iu >> 5), (iv >> 5)], axis=-1).astype(np.int16)
    m2 = ((iv & 31) * Q + (iu & 31)).astype(np.uint16)
    return m1, m2


def invert3x3(m):
    """A.2: cofactor inverse with the documented multiplication order."""
    m = np.asarray(m, np.float64)
    (a, b, c), (d, e, f), (g, h, i) = m
    det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g)
    s = 1.0 / det
    return np.array([[(e * i - f * h) * s, (c * h - b * i) * s, (b * f - c * e) * s],
                     [(f * g - d * i) * s, (a * i - c * g) * s, (c * d - a * f) * s],
                     [(d * h - e * g) * s, (b * g - a * h) * s, (a * e - b * d) * s]])


def perspective_coords(M, dsize):
    """A.2: inverse-map coordinates in Q5, evaluated per 64-column block origin."""
    w, h = dsize
    bw0 = min(1024 // min(16, h), w)
    xs = np.arange(w)
    x0 = (xs // bw0 * bw0).astype(np.float64)[None, :]
    x1 = (xs % bw0).astype(np.float64)[None, :]
    y = np.arange(h, dtype=np.float64)[:, None]
    M = np.asarray(M, np.float64).reshape(3, 3)
    X0 = M[0, 0] * x0 + M[0, 1] * y + M[0, 2]
    Y0 = M[1, 0] * x0 + M[1, 1] * y + M[1, 2]
    W0 = M[2, 0] * x0 + M[2, 1] * y + M[2, 2]
    W = W0 + M[2, 0] * x1
    with np.errstate(divide="ignore"):
        W = np.where(W != 0, Q / W, 0.0)
    lim = lambda a: np.maximum(float(-2 ** 31), np.minimum(float(2 ** 31 - 1), a))
    X = _rne(lim((X0 + M[0, 0] * x1) * W)).astype(np.int64)
    Y = _rne(lim((Y0 + M[1, 0] * x1) * W)).astype(np.int64)
    xy = np.stack([np.clip(X >> 5, -32768, 32767), np.clip(Y >> 5, -32768, 32767)], axis=-1).astype(np.int16)
    a = ((Y & 31) * Q + (X & 31)).astype(np.uint16)
    return xy, a


def _taps(src, sx, sy):
    """Fetch the 2x2 neighbourhood with BORDER_CONSTANT 0 per tap."""
    h, w = src.shape[:2]
    out = []
    for dy in (0, 1):
        for dx in (0, 1):
            xx, yy = sx + dx, sy + dy
            ok = (xx >= 0) & (xx < w) & (yy >= 0) & (yy < h)
            t = src[np.clip(yy, 0, h - 1), np.clip(xx, 0, w - 1)]
            if src.ndim == 3:
                ok = ok[..., None]
            out.append(np.where(ok, t, 0))
    return out


def remap_u8(src, map1, map2):
    """A.4: (sum p*w' + 512) >> 10 with 5-bit-product weights, equal to the Q15 table form for 8-bit data."""
    sx, sy = map1[..., 0].astype(np.int64), map1[..., 1].astype(np.int64)
    code = map2.astype(np.int64) & 1023
    fx, fy = code & 31, code >> 5
    t = [a.astype(np.int64) for a in _taps(src, sx, sy)]
    ws = [(Q - fx) * (Q - fy), fx * (Q - fy), (Q - fx) * fy, fx * fy]
    if src.ndim == 3:
        ws = [x[..., None] for x in ws]
    acc = t[0] * ws[0] + t[1] * ws[1] + t[2] * ws[2] + t[3] * ws[3]
    return ((acc + 512) >> 10).astype(np.uint8)


def remap_f32(src, map1, map2):
    """A.3: float32 weights, float32 left-to-right accumulation, round-half-even, saturate to the source type."""
    sx, sy = map1[..., 0].astype(np.int64), map1[..., 1].astype(np.int64)
    code = map2.astype(np.int64) & 1023
    fx = (code & 31).astype(np.float32) * np.float32(1 / 32)
    fy = (code >> 5).astype(np.float32) * np.float32(1 / 32)
    one = np.float32(1)
    ws = [(one - fy) * (one - fx), (one - fy) * fx, fy * (one - fx), fy * fx]
    t = [a.astype(np.float32) for a in _taps(src, sx, sy)]
    if src.ndim == 3:
        ws = [x[..., None] for x in ws]
    acc = ((t[0] * ws[0] + t[1] * ws[1]) + t[2] * ws[2]) + t[3] * ws[3]
    info = np.iinfo(src.dtype)
    return np.clip(np.rint(acc), info.min, info.max).astype(src.dtype)


def bgr2hsv(img):
    """A.7 forward: integer tables, H in [0, 180)."""
    idx = np.arange(256, dtype=np.float64)
    with np.errstate(divide="ignore"):
        sdiv = np.where(idx > 0, np.rint((255 << 12) / (1.0 * idx)), 0).astype(np.int64)
        hdiv = np.where(idx > 0, np.rint((180 << 12) / (6.0 * idx)), 0).astype(np.int64)
    b, g, r = (img[..., i].astype(np.int64) for i in range(3))
    v = np.maximum(np.maximum(b, g), r)
    diff = v - np.minimum(np.minimum(b, g), r)
    s = (diff * sdiv[v] + 2048) >> 12
    h = np.where(v == r, g - b, np.where(v == g, b - r + 2 * diff, r - g + 4 * diff))
    h = (h * hdiv[diff] + 2048) >> 12
    h = np.where(h < 0, h + 180, h)
    return np.stack([np.clip(h, 0, 255), s, v], axis=-1).astype(np.uint8)


def hsv2bgr(hsv):
    """A.7 inverse: float32 sector formula, *255 and round-half-even."""
    f = np.float32
    h = hsv[..., 0].astype(f) * f(6.0 / 180.0)
    s = hsv[..., 1].astype(f) * f(1.0 / 255.0)
    v = hsv[..., 2].astype(f) * f(1.0 / 255.0)
    h = np.fmod(h, f(6))
    sec = np.floor(h).astype(np.int64)
    frac = h - sec.astype(f)
    bad = (sec < 0) | (sec >= 6)
    sec = np.where(bad, 0, sec)
    frac = np.where(bad, f(0), frac)
    one = f(1)
    tab = np.stack([v, v * (one - s), v * (one - s * frac), v * (one - s * (one - frac))], axis=-1)
    sel = np.array([[1, 3, 0], [1, 0, 2], [3, 0, 1], [0, 2, 1], [0, 1, 3], [2, 1, 0]])[sec]
    bgr = np.take_along_axis(tab, sel, axis=-1)
    bgr = np.where((s == 0)[..., None], v[..., None], bgr)
    return np.clip(np.rint(bgr * f(255)), 0, 255).astype(np.uint8)
