"""ctypes binding of libbevwarp.so (include/bevwarp.h).  No PyTorch, no cv2: numpy arrays in, numpy arrays out.

The library is the ONLY compute path of this package.  If the shared object is missing, or no HIP device is
visible, the calls raise -- there is deliberately no NumPy/CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BEVW_LIB_PATH") or os.path.join(_HERE, "libbevwarp.so")   # override: A/B of two builds
ABI_VERSION = 5

SCHED_AUTO, SCHED_PER_PIXEL, SCHED_TILE_PLAN = 0, 1, 2
PROJ_LUT, PROJ_ANALYTIC, PROJ_ANALYTIC_F32 = 0, 1, 2   # bevw_set_projection
PITCH_DENSE, PITCH_ALIGNED = 0, -1                    # bevw_set_output_pitch
COMPAT_FILLPOLY, COMPAT_ADDWEIGHTED, COMPAT_WARP, COMPAT_REMAP = 0, 1, 2, 3   # bevw_set_compat keys (include/bevwarp.h)


class BevwError(Exception):
    """A libbevwarp call failed (the message is bevw_last_error())."""


class bevw_config(C.Structure):
    _fields_ = [("frame_width", C.c_int32), ("frame_height", C.c_int32), ("bev_width", C.c_int32),
                ("bev_height", C.c_int32), ("car_width", C.c_int32), ("car_height", C.c_int32),
                ("focal_scale", C.c_double), ("size_scale", C.c_double), ("blend", C.c_int32),
                ("balance", C.c_int32), ("device", C.c_int32), ("schedule", C.c_int32)]


# name -> (restype, argtypes); this table is also what tests/test_abi.py checks against include/bevwarp.h
_vp, _i, _sz, _d = C.c_void_p, C.c_int, C.c_size_t, C.c_double
_pvp = C.POINTER(C.c_void_p)
SIGNATURES = {
    "bevw_abi_version": (_i, []),
    "bevw_set_compat": (_i, [_i, _i]),
    "bevw_get_compat": (_i, [_i]),
    "bevw_device_count": (_i, []),
    "bevw_last_error": (C.c_char_p, []),
    "bevw_device_name": (_i, [_i, C.c_char_p, _sz]),
    "bevw_malloc": (_i, [_i, _sz, _pvp]),
    "bevw_free": (_i, [_i, _vp]),
    "bevw_memcpy_h2d": (_i, [_i, _vp, _vp, _sz]),
    "bevw_memcpy_d2h": (_i, [_i, _vp, _vp, _sz]),
    "bevw_memset": (_i, [_i, _vp, _i, _sz]),
    "bevw_device_copy_rate": (_i, [_i, _sz, _i, _i, C.POINTER(C.c_double)]),
    "bevw_create": (_i, [C.POINTER(bevw_config), _pvp]),
    "bevw_set_camera": (_i, [_vp, _i, _vp, _vp, _vp]),
    "bevw_build": (_i, [_vp]),
    "bevw_destroy": (None, [_vp]),
    "bevw_get_undistort_map": (_i, [_vp, _i, _vp, _vp]),
    "bevw_get_lut": (_i, [_vp, _i, _vp, _vp]),
    "bevw_get_mask": (_i, [_vp, _i, _vp]),
    "bevw_plan_info": (_i, [_vp, _vp]),
    "bevw_set_projection": (_i, [_vp, _i]),
    "bevw_run": (_i, [_vp, _vp, _i, _vp, _vp]),
    "bevw_run_device": (_i, [_vp, _vp, _i, _vp, _vp]),
    "bevw_set_output_pitch": (_i, [_vp, _i]),
    "bevw_output_pitch": (_i, [_vp]),
    "bevw_run_cameras": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "bevw_camera_undistort": (_i, [_vp, _i, _vp, _i, _vp]),
    "bevw_camera_warp_homography": (_i, [_vp, _i, _vp, _i, _i, _i, _vp]),
    "bevw_camera_raw2bev": (_i, [_vp, _i, _vp, _i, _vp]),
    "bevw_apply_mask": (_i, [_vp, _i, _vp, _i, _vp]),
    "bevw_set_camera_shard": (_i, [_vp, _vp, _i]),
    "bevw_shard_box": (_i, [_vp, _vp]),
    "bevw_shard_vsums_device": (_i, [_vp, _vp, _i, _vp]),
    "bevw_shard_run_device": (_i, [_vp, _vp, _i, _vp, _vp]),
    "bevw_shard_pack_device": (_i, [_vp, _vp, _i, _vp]),
    "bevw_combine_device": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp]),
    "bevw_comm_available": (_i, []),
    "bevw_comm_unique_id": (_i, [_vp]),
    "bevw_comm_create": (_i, [_i, _i, _i, _vp, _pvp]),
    "bevw_comm_destroy": (None, [_vp]),
    "bevw_shard_allgather_vsums": (_i, [_vp, _vp, _vp, _i, _vp]),
    "bevw_shard_gather_parts": (_i, [_vp, _vp, _vp, _sz, _i, _vp, _vp]),
    "bevw_comm_selftest": (_i, [_vp, _vp, _vp, _vp, _sz]),
    "bevw_luminance_balance": (_i, [_i, _vp, _i, _i, _i, _vp]),
    "bevw_color_balance": (_i, [_i, _vp, _i, _i, _i, _vp]),
    "bevw_sync": (_i, [_vp]),
    "bevw_timer_start": (_i, [_vp]),
    "bevw_timer_stop": (_i, [_vp, C.POINTER(C.c_float)]),
    "bevw_timer_mark": (_i, [_vp, _i]),
    "bevw_timer_between": (_i, [_vp, _i, _i, C.POINTER(C.c_float)]),
    "bevw_fisheye_remapper_create": (_i, [_i, _i, _i, _vp, _vp, _d, _d, _d, _d, _pvp]),
    "bevw_pinhole_remapper_create": (_i, [_i, _i, _i, _vp, _vp, _i, _d, _d, _d, _d, _pvp]),
    "bevw_remapper_from_maps": (_i, [_i, _i, _i, _vp, _vp, _i, _i, _pvp]),
    "bevw_remapper_dims": (_i, [_vp, _vp]),
    "bevw_remapper_get_maps": (_i, [_vp, _vp, _vp]),
    "bevw_remap": (_i, [_vp, _vp, _i, _vp]),
    "bevw_remap_device": (_i, [_vp, _vp, _i, _vp]),
    "bevw_remapper_sync": (_i, [_vp]),
    "bevw_remapper_timer_start": (_i, [_vp]),
    "bevw_remapper_timer_stop": (_i, [_vp, C.POINTER(C.c_float)]),
    "bevw_remapper_timer_mark": (_i, [_vp, _i]),
    "bevw_remapper_timer_between": (_i, [_vp, _i, _i, C.POINTER(C.c_float)]),
    "bevw_remapper_destroy": (None, [_vp]),
    "bevw_warp_perspective_u8c3": (_i, [_i, _vp, _i, _i, _vp, _i, _i, _i, _vp]),
    "bevw_translate_u8c3": (_i, [_i, _vp, _i, _i, _i, _i, _i, _vp]),
    "bevw_resize_dsize": (_i, [_i, _i, _d, _d, _vp]),
    "bevw_resize_linear_u8c3": (_i, [_i, _vp, _i, _i, _d, _d, _i, _vp]),
    "bevw_jpeg_probe": (_i, [_vp, _sz, _vp]),
    "bevw_jpeg_create": (_i, [_i, _pvp]),
    "bevw_jpeg_destroy": (None, [_vp]),
    "bevw_jpeg_decode_stage": (_i, [_vp, _vp, _vp, _i]),
    "bevw_jpeg_decode_run_device": (_i, [_vp, _vp, _sz, _sz]),
    "bevw_jpeg_decode": (_i, [_vp, _vp, _vp, _i, _vp]),
    "bevw_jpeg_decode_info": (_i, [_vp, _vp]),
    "bevw_jpeg_get_planes": (_i, [_vp, _i, _vp]),
    "bevw_jpeg_encode_bound": (_i, [_i, _i, _i, _vp]),
    "bevw_jpeg_encode_run_device": (_i, [_vp, _vp, _i, _i, _i, _sz, _sz, _i, _i]),
    "bevw_jpeg_encoded_sizes": (_i, [_vp, _vp]),
    "bevw_jpeg_encoded_copy": (_i, [_vp, _i, _vp, _sz]),
    "bevw_jpeg_encoded_fetch": (_i, [_vp, _vp, _sz, _vp]),
    "bevw_jpeg_wait_engine": (_i, [_vp, _vp]),
    "bevw_wait_jpeg": (_i, [_vp, _vp]),
    "bevw_jpeg_encode": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "bevw_jpeg_sync": (_i, [_vp]),
    "bevw_jpeg_timer_mark": (_i, [_vp, _i]),
    "bevw_jpeg_timer_between": (_i, [_vp, _i, _i, _vp]),
}

_lib = None


def lib() -> C.CDLL:
    """Load libbevwarp.so (built by cameracalibration_amd/build.py); raises if it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise BevwError(f"{LIB_PATH} is missing: build it with `python -m cameracalibration_amd.build` "
                            "(hipcc, gfx950). There is no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        if L.bevw_abi_version() != ABI_VERSION:
            raise BevwError(f"libbevwarp ABI {L.bevw_abi_version()} != binding ABI {ABI_VERSION}")
        _lib = L
    return _lib


def prefer_hw_queues(n: int = 8) -> bool:
    """Opt-in: ask the HIP runtime for `n` hardware queues (GPU_MAX_HW_QUEUES; 4 by default).  HIP multiplexes a process's streams onto
    them and streams that share a queue do not overlap: the compressed pipeline (BevGenerator.jpeg_stream: an engine stream + three
    codec contexts of two streams each) measured 14.7 k frame sets/s with 4 queues and 16.5 k with 8 (profiles/r04/hw_queues_jpeg_stream.log).
    The variable is read when the HIP runtime initialises, so this must run BEFORE the first libbevwarp call of the process (and before any
    other HIP user); it changes the process environment (child processes inherit it), which is why importing the package does not do it.
    The caller's own setting wins.  Returns True when the setting can still take effect; warns and returns False otherwise."""
    import warnings
    if "GPU_MAX_HW_QUEUES" in os.environ:
        return _lib is None
    if _lib is not None:
        warnings.warn("prefer_hw_queues() after libbevwarp.so was loaded: the HIP runtime has read GPU_MAX_HW_QUEUES already", RuntimeWarning)
        return False
    os.environ["GPU_MAX_HW_QUEUES"] = str(int(n))
    return True


def check(status: int) -> None:
    if status != 0:
        raise BevwError(lib().bevw_last_error().decode("utf-8", "replace") or f"libbevwarp error {status}")


def device_count() -> int:
    return int(lib().bevw_device_count())


def require_device() -> None:
    if device_count() <= 0:
        raise BevwError("no HIP device is visible: cameracalibration_amd has no CPU path")


def device_copy_rate(nbytes: int = 1 << 30, reps: int = 10, streaming: bool = False, device: int = 0) -> float:
    """GB/s a plain copy kernel MOVES (bytes read + bytes written) between two fresh buffers of `nbytes`: a measured yardstick beside the
    8 TB/s specification peak (bevw_device_copy_rate)."""
    g = C.c_double()
    check(lib().bevw_device_copy_rate(int(device), int(nbytes), int(reps), int(bool(streaming)), C.byref(g)))
    return float(g.value)


def device_name(device: int = 0) -> str:
    buf = C.create_string_buffer(256)
    check(lib().bevw_device_name(device, buf, 256))
    return buf.value.decode()


def ptr(a: np.ndarray) -> int:
    if not a.flags["C_CONTIGUOUS"]:
        raise ValueError("array must be C-contiguous")
    return a.ctypes.data


def as_u8_image(img, what="image") -> np.ndarray:
    a = np.ascontiguousarray(img)
    if a.dtype != np.uint8 or a.ndim != 3 or a.shape[2] != 3:
        raise Exception(f"{what} must be uint8 [H, W, 3] (BGR), got {a.dtype} {a.shape}")
    return a


def f64(a, n) -> np.ndarray:
    out = np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(-1)[:n])
    if out.size != n:
        raise Exception(f"expected {n} float64 values, got {out.size}")
    return out


def own_argv(parser, argv=None):
    """The tokens of argv that are EXACTLY one of `parser`'s option strings (plus their value).  The mirror modules parse their
    flags at import time like the reference does (surroundBEV.py:6-17); argparse prefix-matches single-dash options (a foreign
    `-s` is taken for `-ss`) and would abort the host program, so only exact matches are handed to it."""
    import sys

    argv = list(sys.argv[1:] if argv is None else argv)
    opts = parser._option_string_actions
    out, i = [], 0
    while i < len(argv):
        tok = argv[i]
        key = tok.split("=", 1)[0]
        if key in opts:
            out.append(tok)
            if "=" not in tok and opts[key].nargs != 0 and i + 1 < len(argv):
                out.append(argv[i + 1])
                i += 1
        i += 1
    return out


class DeviceBuffer:
    """A raw HBM allocation (bevw_malloc) -- lets bench.py / tests keep batches resident without a GPU framework."""

    def __init__(self, nbytes: int, device: int = 0):
        self.device, self.nbytes = device, int(nbytes)
        p = C.c_void_p()
        check(lib().bevw_malloc(device, self.nbytes, C.byref(p)))
        self.ptr = p.value

    def upload(self, arr: np.ndarray, offset: int = 0) -> "DeviceBuffer":
        arr = np.ascontiguousarray(arr)
        assert offset + arr.nbytes <= self.nbytes
        check(lib().bevw_memcpy_h2d(self.device, self.ptr + offset, ptr(arr), arr.nbytes))
        return self

    def download(self, shape, dtype=np.uint8, offset: int = 0) -> np.ndarray:
        out = np.empty(shape, dtype)
        assert offset + out.nbytes <= self.nbytes
        check(lib().bevw_memcpy_d2h(self.device, ptr(out), self.ptr + offset, out.nbytes))
        return out

    def fill(self, value: int = 0) -> None:
        check(lib().bevw_memset(self.device, self.ptr, value, self.nbytes))

    def free(self) -> None:
        if self.ptr:
            lib().bevw_free(self.device, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
