"""ctypes front-end of the JPEG oracle (oracle/jpegoracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.  The restatement is PINNED:
tests/test_jpeg_oracle.py holds it against Pillow's libjpeg-turbo (the library cv2.imread / cv2.imwrite wrap, main.py:74-77,
surroundBEV.py:340) byte for byte, decode and encode.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libjpegoracle.so")
_lib = None

SAMPLING_420, SAMPLING_422, SAMPLING_444 = 0x22, 0x21, 0x11


def build(force: bool = False) -> str:
    if os.environ.get("BEVW_ORACLE_SANITIZE") == "1":   # ASan + UBSan build (tests/test_sanitizers.py); see oracle.sanitized_build
        from oracle import oracle as _O
        return _O.sanitized_build("jpegoracle")
    src = os.path.join(_HERE, "jpegoracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "-s", "libjpegoracle.so"] + (["-B"] if force else []), check=True)
    return _LIB_PATH


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        L.jo_probe.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_int32)]
        L.jo_decode_bgr.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_void_p]
        L.jo_encode_bound.argtypes = [C.c_int, C.c_int]
        L.jo_encode_bound.restype = C.c_size_t
        L.jo_encode_bgr.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
        L.jo_encode_bgr.restype = C.c_long
        _lib = L
    return _lib


def probe(raw: bytes) -> dict:
    info = (C.c_int32 * 8)()
    s = lib().jo_probe(raw, len(raw), info)
    if s:
        raise ValueError(f"jo_probe: {s}")
    return dict(width=info[0], height=info[1], components=info[2], h_samp=info[3], v_samp=info[4], restart_interval=info[5],
                orientation=info[6])


def plane_shapes(info: dict):
    """[(rows, cols)] of the sample planes jo_decode_bgr can hand back (whole MCUs)."""
    nc = info["components"]
    hs, vs = (info["h_samp"], info["v_samp"]) if nc == 3 else (1, 1)
    mcux = -(-info["width"] // (8 * hs))
    mcuy = -(-info["height"] // (8 * vs))
    shapes = [(mcuy * vs * 8, mcux * hs * 8)]
    if nc == 3:
        shapes += [(mcuy * 8, mcux * 8)] * 2
    return shapes


def imdecode(raw: bytes, planes: bool = False):
    """cv2.imread / cv2.imdecode(..., IMREAD_COLOR) of a JPEG: BGR uint8 [h][w][3] (and the sample planes after the IDCT)."""
    info = probe(raw)
    out = np.empty((info["height"], info["width"], 3), np.uint8)
    shapes = plane_shapes(info)
    pl = np.empty(sum(r * c for r, c in shapes), np.uint8) if planes else None
    s = lib().jo_decode_bgr(raw, len(raw), out.ctypes.data, pl.ctypes.data if planes else None)
    if s:
        raise ValueError(f"jo_decode_bgr: {s}")
    if not planes:
        return out
    parts, o = [], 0
    for r, c in shapes:
        parts.append(pl[o:o + r * c].reshape(r, c))
        o += r * c
    return out, parts


def imencode(bgr: np.ndarray, quality: int = 95, sampling: int = SAMPLING_420) -> bytes:
    """cv2.imwrite('x.jpg', bgr) / cv2.imencode('.jpg', bgr): the file libjpeg writes at `quality` (cv2's default 95), 4:2:0."""
    bgr = np.ascontiguousarray(bgr, np.uint8)
    h, w = bgr.shape[:2]
    cap = lib().jo_encode_bound(w, h)
    out = np.empty(cap, np.uint8)
    n = lib().jo_encode_bgr(bgr.ctypes.data, w, h, w * 3, quality, sampling, out.ctypes.data, cap)
    if n <= 0:
        raise ValueError(f"jo_encode_bgr: {n}")
    return out[:n].tobytes()
