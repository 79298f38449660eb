"""Host-side mirror of the hot-path slice of ExtrinsicCalibration/extrinsicCalib.py: ExCalibrator.warp
(extrinsicCalib.py:166-169) = cv2.warpPerspective(src_img, homography, (dst_w, dst_h)) on the GPU.

Homography ESTIMATION (chessboard corners + cv2.findHomography RANSAC, extrinsicCalib.py:171-183) is out of scope
(SURVEY.md section 2 row 10): the homography comes from `set_homography(H, src_img, dst_img)` or a saved
camera_<id>_H.npy (extrinsicCalib.py:216).
"""
from __future__ import annotations

import argparse

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

parser = argparse.ArgumentParser(description="Homography from Source to Destination Image")
parser.add_argument('-id', '--CAMERA_ID', default=1, type=int, help='Camera ID')
parser.add_argument('-path', '--INPUT_PATH', default='./data/', type=str, help='Input Source/Destination Image Path')
args, _unknown = parser.parse_known_args()


class ExCalibrator:
    """extrinsicCalib.py:132-183 -- get_args() and warp() keep the reference's behaviour."""

    def __init__(self, device: int = 0):
        self.src_corners_total = np.empty((0, 1, 2))
        self.dst_corners_total = np.empty((0, 1, 2))
        self.homography = None
        self.src_img = None
        self.dst_img = None
        self.device = device

    @staticmethod
    def get_args():
        return args

    def set_homography(self, homography, src_img, dst_img):
        """Additive: what __call__ leaves behind (extrinsicCalib.py:180-183) when the homography is already known.
        dst_img may also be a (height, width) pair."""
        self.homography = np.array(homography, dtype=np.float64).reshape(3, 3)
        self.src_img = _ffi.as_u8_image(src_img, "src_img")
        self.dst_img = dst_img
        return self.homography

    def warp(self):
        """extrinsicCalib.py:166-169"""
        if self.homography is None or self.src_img is None:
            raise Exception("no homography: call set_homography(H, src_img, dst_img) first")
        _ffi.require_device()
        shape = self.dst_img.shape if hasattr(self.dst_img, "shape") else tuple(self.dst_img)
        dh, dw = int(shape[0]), int(shape[1])
        src = self.src_img
        out = np.empty((dh, dw, 3), np.uint8)
        check(lib().bevw_warp_perspective_u8c3(self.device, ptr(src), src.shape[1], src.shape[0],
                                               ptr(f64(self.homography, 9)), dw, dh, 1, ptr(out)))
        return out

    def __call__(self, src_img, dst_img):
        raise Exception("homography estimation is out of scope of cameracalibration_amd (use set_homography)")
