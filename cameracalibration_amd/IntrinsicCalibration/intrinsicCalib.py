"""Host-side mirror of the hot-path slice of IntrinsicCalibration/intrinsicCalib.py: InCalibrator.undistort
(intrinsicCalib.py:193-195) = cv2.remap through the fisheye undistort maps of intrinsicCalib.py:90-103, on the GPU.

The calibration SOLVER (chessboard detection, cv2.fisheye.calibrate, intrinsicCalib.py:62-88,179-191) is out of
scope of this engine: K and D come from `set_calibration(K, D)` or from the
camera_<id>_{K,D}.npy files the reference's main() writes (intrinsicCalib.py:413-414).
"""
from __future__ import annotations

import argparse
import ctypes as C

import numpy as np

try:
    from .. import _ffi
except ImportError:
    # imported as a TOP-LEVEL package (sys.path points inside cameracalibration_amd/, the drop-in layout of main.py:5-7)
    import importlib
    import os as _os
    import sys as _sys

    _sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))))
    _ffi = importlib.import_module("cameracalibration_amd._ffi")
check, f64, lib, ptr = _ffi.check, _ffi.f64, _ffi.lib, _ffi.ptr

# the flags of intrinsicCalib.py:6-29 that shape the undistort maps (same names / defaults)
parser = argparse.ArgumentParser(description="Camera Intrinsic Calibration")
parser.add_argument('-type', '--CAMERA_TYPE', default='fisheye', type=str, help='Camera Type: fisheye/normal')
parser.add_argument('-id', '--CAMERA_ID', default=1, type=int, help='Camera ID')
parser.add_argument('-fw', '--FRAME_WIDTH', default=1280, type=int, help='Camera Frame Width')
parser.add_argument('-fh', '--FRAME_HEIGHT', default=1024, type=int, help='Camera Frame Height')
parser.add_argument('-fs', '--FOCAL_SCALE', default=0.5, type=float, help='Camera Undistort Focal Scale')
parser.add_argument('-ss', '--SIZE_SCALE', default=1, type=float, help='Camera Undistort Size Scale')
args, _unknown = parser.parse_known_args(_ffi.own_argv(parser))


class CalibData:
    """intrinsicCalib.py:32-42 (the fields the undistort path uses)."""

    def __init__(self):
        self.type = None
        self.camera_mat = None
        self.dist_coeff = None
        self.map1 = None
        self.map2 = None
        self.ok = False


class Fisheye:
    def __init__(self):
        self.data = CalibData()
        self.data.type = "FISHEYE"
        self._remapper = None
        self._device = 0

    def _get_camera_mat_dst(self, camera_mat):
        """intrinsicCalib.py:90-96"""
        camera_mat_dst = camera_mat.copy()
        camera_mat_dst[0][0] *= args.FOCAL_SCALE
        camera_mat_dst[1][1] *= args.FOCAL_SCALE
        camera_mat_dst[0][2] = args.FRAME_WIDTH / 2 * args.SIZE_SCALE
        camera_mat_dst[1][2] = args.FRAME_HEIGHT / 2 * args.SIZE_SCALE
        return camera_mat_dst

    def _release(self):
        if self._remapper:
            lib().bevw_remapper_destroy(self._remapper)
            self._remapper = None

    def _get_undistort_maps(self):
        """intrinsicCalib.py:98-103 -- maps are built by k_fisheye_map and stay on the device."""
        _ffi.require_device()
        self._release()
        d = self.data
        r = C.c_void_p()
        check(lib().bevw_fisheye_remapper_create(self._device, int(args.FRAME_WIDTH), int(args.FRAME_HEIGHT),
                                                 ptr(f64(d.camera_mat, 9)), ptr(f64(d.dist_coeff, 4)),
                                                 float(args.FOCAL_SCALE), float(args.SIZE_SCALE), 0.0, 0.0, C.byref(r)))
        self._remapper = r
        dims = np.zeros(4, np.int32)
        check(lib().bevw_remapper_dims(r, ptr(dims)))
        d.map1 = np.empty((dims[3], dims[2], 2), np.int16)
        d.map2 = np.empty((dims[3], dims[2]), np.uint16)
        check(lib().bevw_remapper_get_maps(r, ptr(d.map1), ptr(d.map2)))
        d.ok = True

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass


class Normal(Fisheye):
    """intrinsicCalib.py:108-163 -- pinhole model: cv2.initUndistortRectifyMap with D = k1 k2 p1 p2 k3 (k_pinhole_map)."""

    def __init__(self):
        super().__init__()
        self.data.type = "NORMAL"

    def _get_undistort_maps(self):
        """intrinsicCalib.py:158-163"""
        _ffi.require_device()
        self._release()
        d = self.data
        dist = np.ascontiguousarray(np.asarray(d.dist_coeff, np.float64).reshape(-1))
        r = C.c_void_p()
        check(lib().bevw_pinhole_remapper_create(self._device, int(args.FRAME_WIDTH), int(args.FRAME_HEIGHT),
                                                 ptr(f64(d.camera_mat, 9)), ptr(dist), int(dist.size),
                                                 float(args.FOCAL_SCALE), float(args.SIZE_SCALE), 0.0, 0.0, C.byref(r)))
        self._remapper = r
        dims = np.zeros(4, np.int32)
        check(lib().bevw_remapper_dims(r, ptr(dims)))
        d.map1 = np.empty((dims[3], dims[2], 2), np.int16)
        d.map2 = np.empty((dims[3], dims[2]), np.uint16)
        check(lib().bevw_remapper_get_maps(r, ptr(d.map1), ptr(d.map2)))
        d.ok = True


class InCalibrator:
    """intrinsicCalib.py:165-224 -- construction, get_args() and undistort() keep the reference's behaviour."""

    def __init__(self, camera):
        if camera == 'fisheye':
            self.camera = Fisheye()
        elif camera == 'normal':
            self.camera = Normal()
        else:
            raise Exception("camera should be fisheye/normal")
        self.corners = []

    @staticmethod
    def get_args():
        return args

    def set_calibration(self, camera_mat, dist_coeff):
        """Additive: install K (3x3) and D (4) and build the undistort maps (what Fisheye.update ends with,
        intrinsicCalib.py:52-60)."""
        d = self.camera.data
        d.camera_mat = np.array(camera_mat, dtype=np.float64).reshape(3, 3)
        d.dist_coeff = np.array(dist_coeff, dtype=np.float64).reshape(-1, 1)
        self.camera._get_undistort_maps()
        return d

    def undistort(self, img):
        """intrinsicCalib.py:193-195: cv2.remap(img, data.map1, data.map2, cv2.INTER_LINEAR)."""
        cam = self.camera
        if cam._remapper is None:
            raise Exception("no calibration: call set_calibration(K, D) first")
        img = _ffi.as_u8_image(img)
        if img.shape[:2] != (args.FRAME_HEIGHT, args.FRAME_WIDTH):
            raise Exception("image is {}x{}, FRAME is {}x{}".format(img.shape[1], img.shape[0], args.FRAME_WIDTH,
                                                                   args.FRAME_HEIGHT))
        h, w = cam.data.map2.shape
        out = np.empty((h, w, 3), np.uint8)
        check(lib().bevw_remap(cam._remapper, ptr(img), 1, ptr(out)))
        return out

    def undistort_batch(self, imgs):
        """Additive: uint8 [B, FH, FW, 3] -> uint8 [B, h, w, 3]."""
        cam = self.camera
        if cam._remapper is None:
            raise Exception("no calibration: call set_calibration(K, D) first")
        imgs = np.ascontiguousarray(imgs)
        if imgs.dtype != np.uint8 or imgs.ndim != 4 or imgs.shape[1:] != (args.FRAME_HEIGHT, args.FRAME_WIDTH, 3):
            raise Exception("images must be uint8 [B, {}, {}, 3]".format(args.FRAME_HEIGHT, args.FRAME_WIDTH))
        h, w = cam.data.map2.shape
        out = np.empty((imgs.shape[0], h, w, 3), np.uint8)
        check(lib().bevw_remap(cam._remapper, ptr(imgs), imgs.shape[0], ptr(out)))
        return out

    def calibrate(self, img):
        raise Exception("the calibration solver is out of scope of cameracalibration_amd (use set_calibration)")

    def __call__(self, raw_frame):
        raise Exception("the calibration solver is out of scope of cameracalibration_amd (use set_calibration)")


class CalibMode:
    """intrinsicCalib.py:226-396 -- the reference's capture / calibration DRIVER (camera, video or image input feeding
    chessboard frames to ``InCalibrator.__call__`` with cv2 GUI windows).  It is exported so that ``main.py:5``
    (``from IntrinsicCalibration import InCalibrator, CalibMode``) imports unchanged; running it needs the calibration solver
    and the cv2 GUI, both outside this engine's scope (iterative LM solver + corner
    detector, offline, not data-parallel).  Calibrate with the reference, then hand K and D to ``set_calibration`` --
    or load the ``camera_<id>_{K,D}.npy`` files its ``main()`` writes (intrinsicCalib.py:413-414)."""

    def __init__(self, calibrator, input_type, mode):
        self.calibrator = calibrator
        self.input_type = input_type
        self.mode = mode

    def __call__(self):
        raise Exception("CalibMode drives the chessboard calibration solver and the cv2 GUI, which are out of scope of "
                        "cameracalibration_amd: calibrate with the reference and use InCalibrator.set_calibration(K, D)")

