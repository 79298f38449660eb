"""Host-side mirror of the hot-path slice of ExtrinsicCalibration/extrinsicCalib.py: ExCalibrator.warp
(extrinsicCalib.py:166-169) = cv2.warpPerspective(src_img, homography, (dst_w, dst_h)) on the GPU, and the two
pre-processing warps CenterImage.translate (:54-59, cv2.warpAffine) and ScaleImage.__call__ (:122-130, cv2.resize +
pad / centre-crop).  The mouse / window parts of CenterImage are GUI code and stay out.

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
parser.add_argument('-bw', '--BORAD_WIDTH', default=7, type=int, help='Chess Board Width (corners number)')
parser.add_argument('-bh', '--BORAD_HEIGHT', default=6, type=int, help='Chess Board Height (corners number)')
parser.add_argument('-size', '--SCALED_SIZE', default=10, type=int, help='Scaled Chess Board Square Size (image pixel)')
args, _unknown = parser.parse_known_args(_ffi.own_argv(parser))


class CenterImage:
    """extrinsicCalib.py:22-85 without the window / mouse loop: the picked point is set with `set_center(x, y)`."""

    def __init__(self, device: int = 0):
        self.x = 0
        self.y = 0
        self.device = device

    def set_center(self, x, y):
        self.x, self.y = int(x), int(y)
        return self

    def translate(self, img):
        """extrinsicCalib.py:54-59"""
        _ffi.require_device()
        img = _ffi.as_u8_image(img)
        shift_x = img.shape[1] // 2 - self.x
        shift_y = img.shape[0] // 2 - self.y
        out = np.empty_like(img)
        check(lib().bevw_translate_u8c3(self.device, ptr(img), img.shape[1], img.shape[0], shift_x, shift_y, 1, ptr(out)))
        return out

    def __call__(self, raw_frame):
        """extrinsicCalib.py:61-85 after the user has confirmed a point: (0, 0) means "keep the frame"."""
        if not (self.x == 0 and self.y == 0):
            return self.translate(raw_frame)
        return raw_frame


class ScaleImage:
    """extrinsicCalib.py:87-130.  `corners`: the chessboard corners array [BORAD_WIDTH * BORAD_HEIGHT, 2] (or
    [.., 1, 2] as cv2.findChessboardCorners returns it)."""

    def __init__(self, corners, device: int = 0):
        self.device = device
        self.calc_dist(corners)
        print('scale image from {} to {}'.format(self.dist_square, args.SCALED_SIZE))
        self.scale_factor = args.SCALED_SIZE / self.dist_square

    def calc_dist(self, corners):
        c = np.asarray(corners, dtype=np.float64).reshape(-1, 2)
        dist_total = 0
        for i in range(args.BORAD_HEIGHT):
            a, b = c[i * args.BORAD_WIDTH], c[(i + 1) * args.BORAD_WIDTH - 1]
            dist = float(np.sqrt(((a - b) ** 2).sum()))     # cv2.norm(a, b, cv2.NORM_L2)
            dist_total += dist / (args.BORAD_WIDTH - 1)
        self.dist_square = dist_total / args.BORAD_HEIGHT

    def padding(self, img, width, height):
        H, W = img.shape[0], img.shape[1]
        top = (height - H) // 2
        left = (width - W) // 2
        out = np.zeros((height, width, 3), np.uint8)      # cv2.copyMakeBorder(..., BORDER_CONSTANT, value=(0,0,0))
        out[top:top + H, left:left + W] = img
        return out

    def center_crop(self, img, width, height):
        H, W = img.shape[0], img.shape[1]
        top = (H - height) // 2
        left = (W - width) // 2
        return img[top:top + height, left:left + width]

    def resize(self, raw_frame):
        """cv2.resize(raw_frame, (0,0), fx=scale_factor, fy=scale_factor) (extrinsicCalib.py:125)"""
        _ffi.require_device()
        img = _ffi.as_u8_image(raw_frame)
        ds = np.zeros(2, np.int32)
        f = float(self.scale_factor)
        check(lib().bevw_resize_dsize(img.shape[1], img.shape[0], f, f, ptr(ds)))
        out = np.empty((int(ds[1]), int(ds[0]), 3), np.uint8)
        check(lib().bevw_resize_linear_u8c3(self.device, ptr(img), img.shape[1], img.shape[0], f, f, 1, ptr(out)))
        return out

    def __call__(self, raw_frame):
        width = raw_frame.shape[1]
        height = raw_frame.shape[0]
        raw_frame = self.resize(raw_frame)
        if self.scale_factor < 1:
            raw_frame = self.padding(raw_frame, width, height)
        else:
            raw_frame = self.center_crop(raw_frame, width, height)
        return raw_frame


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
