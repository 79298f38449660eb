"""Drop-in for the hot-path part of the reference's ExtrinsicCalibration package (ExCalibrator.warp)."""
from .extrinsicCalib import ExCalibrator  # noqa: F401
