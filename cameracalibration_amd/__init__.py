"""cameracalibration_amd -- MI355X-native surround-BEV warping engine.

Drop-in for the per-pixel hot path of dyfcalid/CameraCalibration (BevGenerator.__call__, Camera.undistort /
raw2bev / warp_homography, InCalibrator.undistort, ExCalibrator.warp): hand-written HIP kernels for gfx950 behind a
C-ABI shared library (include/bevwarp.h), called from Python through ctypes.  See DESIGN.md / INTEGRATION.md.
"""
from ._ffi import BevwError, DeviceBuffer, device_count, device_name, lib, require_device  # noqa: F401

__all__ = ["BevwError", "DeviceBuffer", "device_count", "device_name", "lib", "require_device"]
