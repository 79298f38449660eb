"""Drop-in for the hot-path part of the reference's IntrinsicCalibration package (InCalibrator.undistort); the names
main.py:5 imports exist, the calibration solver / capture driver raise (out of scope, see intrinsicCalib.py)."""
from .intrinsicCalib import InCalibrator, CalibMode  # noqa: F401
