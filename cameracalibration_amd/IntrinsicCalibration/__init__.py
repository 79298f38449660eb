"""Drop-in for the hot-path part of the reference's IntrinsicCalibration package (InCalibrator.undistort)."""
from .intrinsicCalib import InCalibrator  # noqa: F401
