"""JPEG either side of the path on the GPU -- the cv2.imgcodecs calls of the reference.

The reference reads camera frames with ``cv2.imread`` (main.py:74-77, Tools/undistort.py:65, extrinsicCalib.py:203-204) and writes
its results with ``cv2.imwrite`` (SurroundBirdEyeView/surroundBEV.py:340, Tools/undistort.py:73, extrinsicCalib.py:211).  This module keeps
those names -- ``imread / imwrite / imdecode / imencode`` -- for JPEG files and adds the batch form the engine is for
(:class:`JpegCodec`): many files of one geometry decoded straight into the frame-set layout ``BevGenerator.run_device`` reads, and
device images encoded into complete ``.jpg`` files, without the pixels ever visiting the host.

Everything is computed by libbevwarp's HIP kernels (include/bevwarp.h, ``bevw_jpeg_*``), bit-exact against libjpeg-turbo, the library
behind cv2's JPEG codec.  There is no CPU decoder here: files outside the supported subset (progressive, arithmetic, CMYK, 12-bit,
multi-scan) and non-JPEG files raise.  EXIF orientations are applied on the GPU, as cv2.imread applies them.
"""
from __future__ import annotations

import ctypes as C
from typing import Sequence

import numpy as np

from . import _ffi
from ._ffi import check, lib, ptr

IMREAD_COLOR = 1
IMWRITE_JPEG_QUALITY = 1            # cv2's flag value
SAMPLING_420, SAMPLING_422, SAMPLING_444 = 0x22, 0x21, 0x11
_DEFAULT_QUALITY = 95               # cv2.imwrite's default for .jpg


def probe(raw: bytes) -> dict:
    """Header of a JPEG file: size AS cv2.imread RETURNS IT (EXIF orientation applied; stored_width / stored_height are the frame header's),
    components, luma sampling, restart interval, EXIF orientation.  Raises on unsupported files."""
    info = (C.c_int32 * 8)()
    raw = bytes(raw)
    check(lib().bevw_jpeg_probe(raw, len(raw), info))
    d = dict(width=info[0], height=info[1], components=info[2], h_samp=info[3], v_samp=info[4], restart_interval=info[5],
             orientation=info[6], stored_width=info[0], stored_height=info[1])
    if d["orientation"] >= 5:   # cv2.imread applies the EXIF orientation: 5 .. 8 turn the image by 90 degrees
        d["width"], d["height"] = d["height"], d["width"]
    return d


class JpegCodec:
    """A decode / encode context on one device (``bevw_jpeg``): its own HIP stream, staging memory and scratch.

    ``decode`` / ``encode`` are the host-array forms; ``decode_stage`` + ``decode_run_device`` and ``encode_run_device`` + ``files``
    are the resident forms (raw device pointers, e.g. ``DeviceBuffer.ptr``)."""

    def __init__(self, device: int = 0):
        _ffi.require_device()
        self.device = device
        h = C.c_void_p()
        check(lib().bevw_jpeg_create(device, C.byref(h)))
        self.h = h
        self._n_enc = 0

    def close(self) -> None:
        if getattr(self, "h", None):
            lib().bevw_jpeg_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ---- decode --------------------------------------------------------------------------------------------------
    def decode_stage(self, files: Sequence[bytes]) -> dict:
        """Parse the headers, copy the entropy-coded bytes to pinned memory, enqueue their upload (one geometry per call).  Returns the probe
        of the first file."""
        files = [bytes(f) for f in files]
        if not files:
            raise Exception("no files")
        n = len(files)
        arr = (C.c_char_p * n)(*files)
        lens = (C.c_size_t * n)(*[len(f) for f in files])
        check(lib().bevw_jpeg_decode_stage(self.h, arr, lens, n))
        self._staged = files   # keeps the bytes alive while the library reads them (it has copied them when the call returns)
        return probe(files[0])

    def decode_run_device(self, d_out: int, image_stride_bytes: int, row_pitch_bytes: int) -> None:
        check(lib().bevw_jpeg_decode_run_device(self.h, d_out, image_stride_bytes, row_pitch_bytes))

    def decode(self, files: Sequence[bytes]) -> np.ndarray:
        """``[cv2.imread(f) for f in files]`` as one array uint8 [n, h, w, 3] (BGR)."""
        info = self.decode_stage(files)
        n = len(files)
        out = np.empty((n, info["height"], info["width"], 3), np.uint8)
        check(lib().bevw_jpeg_decode_run_device(self.h, self._host_target(out.nbytes), out[0].nbytes, info["width"] * 3))
        check(lib().bevw_jpeg_sync(self.h))
        short = self.decode_info()["short_images"]
        if short:
            raise _ffi.BevwError(f"{short} of the {n} files end before their image is complete (truncated / corrupt entropy-coded data)")
        check(lib().bevw_memcpy_d2h(self.device, ptr(out), self._d_tmp.ptr, out.nbytes))
        return out

    def _host_target(self, nbytes: int) -> int:
        if getattr(self, "_d_tmp", None) is None or self._d_tmp.nbytes < nbytes:
            if getattr(self, "_d_tmp", None) is not None:
                self._d_tmp.free()
            self._d_tmp = _ffi.DeviceBuffer(nbytes, self.device)
        return self._d_tmp.ptr

    def decode_info(self) -> dict:
        info = (C.c_int64 * 8)()
        check(lib().bevw_jpeg_decode_info(self.h, info))
        return dict(images=info[0], width=info[1], height=info[2], subsequences=info[3], rounds=info[4], entropy_bytes=info[5],
                    short_images=info[6], table_sets=info[7])

    def planes(self, index: int, nbytes: int) -> np.ndarray:
        out = np.empty(nbytes, np.uint8)
        check(lib().bevw_jpeg_get_planes(self.h, index, ptr(out)))
        return out

    # ---- encode --------------------------------------------------------------------------------------------------
    def encode_run_device(self, d_bgr: int, n: int, width: int, height: int, image_stride_bytes: int, row_pitch_bytes: int,
                          quality: int = _DEFAULT_QUALITY, sampling: int = SAMPLING_420) -> None:
        check(lib().bevw_jpeg_encode_run_device(self.h, d_bgr, n, width, height, image_stride_bytes, row_pitch_bytes, quality, sampling))
        self._n_enc = n

    def files(self, copy: bool = True) -> list:
        """The files of the last ``encode_run_device`` (synchronises): one gather on the device, ONE device-to-host copy
        (``bevw_jpeg_encoded_fetch``).  ``copy=True``: a list of ``bytes``; ``copy=False``: ``memoryview`` slices of one host buffer that
        belongs to the returned views (no per-file copy on the host)."""
        n = self._n_enc
        sizes = (C.c_size_t * n)()
        check(lib().bevw_jpeg_encoded_sizes(self.h, sizes))       # (waits for the encode; the sizes are cached for the fetch)
        total = sum(sizes)
        buf = np.empty(total, np.uint8)
        offsets = (C.c_size_t * (n + 1))()
        check(lib().bevw_jpeg_encoded_fetch(self.h, ptr(buf), buf.nbytes, offsets))
        view = memoryview(buf)
        if copy:
            return [view[offsets[i]:offsets[i + 1]].tobytes() for i in range(n)]
        return [view[offsets[i]:offsets[i + 1]] for i in range(n)]

    def encode(self, images, quality: int = _DEFAULT_QUALITY, sampling: int = SAMPLING_420) -> list:
        """``[cv2.imencode('.jpg', im)[1].tobytes() for im in images]``: images uint8 [n, h, w, 3] (BGR) -> complete files."""
        images = np.ascontiguousarray(images)
        if images.dtype != np.uint8 or images.ndim != 4 or images.shape[3] != 3:
            raise Exception("images must be uint8 [n, h, w, 3] (BGR)")
        n, h, w = images.shape[:3]
        d = self._host_target(images.nbytes)
        check(lib().bevw_memcpy_h2d(self.device, d, ptr(images), images.nbytes))
        self.encode_run_device(d, n, w, h, h * w * 3, w * 3, quality, sampling)
        return self.files()

    # ---- stream ---------------------------------------------------------------------------------------------------
    def wait_engine(self, engine_handle) -> None:
        """What is enqueued on this context afterwards starts when everything enqueued on the engine (a ``bevw_handle``) so far is done."""
        check(lib().bevw_jpeg_wait_engine(self.h, engine_handle))

    def engine_waits(self, engine_handle) -> None:
        """The engine's stream waits for everything enqueued on this context so far (no host synchronisation)."""
        check(lib().bevw_wait_jpeg(engine_handle, self.h))

    def sync(self) -> None:
        check(lib().bevw_jpeg_sync(self.h))

    def timer_mark(self, slot: int) -> None:
        check(lib().bevw_jpeg_timer_mark(self.h, int(slot)))

    def timer_between(self, a: int, b: int) -> float:
        ms = C.c_float()
        check(lib().bevw_jpeg_timer_between(self.h, int(a), int(b), C.byref(ms)))
        return float(ms.value)


_codecs = {}


def _codec(device: int = 0) -> JpegCodec:
    if device not in _codecs:
        _codecs[device] = JpegCodec(device)
    return _codecs[device]


def _quality(params) -> int:
    q = _DEFAULT_QUALITY
    if params:
        p = list(params)
        for k, v in zip(p[0::2], p[1::2]):
            if int(k) == IMWRITE_JPEG_QUALITY:
                q = int(v)
    return min(100, max(0, q)) or 1


def imdecode(buf, flags: int = IMREAD_COLOR, device: int = 0) -> np.ndarray:
    """cv2.imdecode(buf, cv2.IMREAD_COLOR) for JPEG data: BGR uint8 [h, w, 3]."""
    if flags != IMREAD_COLOR:
        raise Exception("only IMREAD_COLOR is supported")
    raw = buf.tobytes() if isinstance(buf, np.ndarray) else bytes(buf)
    return _codec(device).decode([raw])[0]


def imread(path: str, flags: int = IMREAD_COLOR, device: int = 0) -> np.ndarray:
    """cv2.imread(path) for .jpg files (main.py:74-77)."""
    with open(path, "rb") as f:
        return imdecode(f.read(), flags, device)


def imencode(ext: str, img, params=None, device: int = 0):
    """cv2.imencode('.jpg', img[, [cv2.IMWRITE_JPEG_QUALITY, q]]) -> (True, uint8 array)."""
    if ext.lower() not in (".jpg", ".jpeg"):
        raise Exception("only .jpg is supported")
    img = _ffi.as_u8_image(img)
    data = _codec(device).encode(img[None], _quality(params))[0]
    return True, np.frombuffer(data, np.uint8)


def imwrite(path: str, img, params=None, device: int = 0) -> bool:
    """cv2.imwrite(path.jpg, img) (surroundBEV.py:340, Tools/undistort.py:73): libjpeg's file at quality 95, 4:2:0."""
    ext = "." + path.rsplit(".", 1)[-1] if "." in path else ""
    ok, data = imencode(ext, img, params, device)
    with open(path, "wb") as f:
        f.write(data.tobytes())
    return ok
