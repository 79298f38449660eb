"""The C-ABI library loads on a GPU-less host and exports every symbol include/bevwarp.h declares; the product path
fails LOUDLY (no CPU fallback) when no HIP device is visible.  No compute calls here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def ffi():
    from cameracalibration_amd import _ffi, build

    build.build()
    return _ffi


def header_symbols():
    text = open(os.path.join(ROOT, "include", "bevwarp.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bevw_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree(ffi):
    syms = header_symbols()
    assert len(syms) >= 35
    assert sorted(ffi.SIGNATURES) == syms


def test_library_exports_every_declared_symbol(ffi):
    L = C.CDLL(ffi.LIB_PATH)
    for name in header_symbols():
        assert hasattr(L, name), name
    assert ffi.lib().bevw_abi_version() == ffi.ABI_VERSION


def test_config_struct_layout(ffi):
    # bevw_config in the header: 6 x int32, 2 x double, 4 x int32
    assert C.sizeof(ffi.bevw_config) == 6 * 4 + 2 * 8 + 4 * 4
    assert ffi.bevw_config.focal_scale.offset == 24 and ffi.bevw_config.blend.offset == 40


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="a GPU is present: the no-device path is not reachable")
def test_fails_loudly_without_a_device(ffi):
    assert ffi.device_count() == 0
    with pytest.raises(ffi.BevwError, match="no HIP device|CPU"):
        ffi.require_device()
    cfg = ffi.bevw_config(64, 48, 40, 40, 10, 16, 1.0, 2.0, 0, 0, 0, 0)
    h = C.c_void_p()
    assert ffi.lib().bevw_create(C.byref(cfg), C.byref(h)) == -2  # BEVW_E_NO_DEVICE
    assert b"no HIP device" in ffi.lib().bevw_last_error()
    p = C.c_void_p()
    assert ffi.lib().bevw_malloc(0, 16, C.byref(p)) == -2
    img = np.zeros((4, 4, 3), np.uint8)
    assert ffi.lib().bevw_color_balance(0, img.ctypes.data, 1, 4, 4, img.ctypes.data) == -2


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="a GPU is present")
def test_python_surface_raises_without_a_device(ffi):
    from cameracalibration_amd import workloads as W
    from cameracalibration_amd.SurroundBirdEyeView import surroundBEV as SB
    from cameracalibration_amd.IntrinsicCalibration import InCalibrator
    from cameracalibration_amd.ExtrinsicCalibration import ExCalibrator

    with pytest.raises(ffi.BevwError):
        SB.BevGenerator(rig=W.repo_rig())
    cal = InCalibrator("fisheye")
    K, D = W.undistort_calibration()
    with pytest.raises(ffi.BevwError):
        cal.set_calibration(K, D)
    ex = ExCalibrator()
    ex.set_homography(np.eye(3), np.zeros((8, 8, 3), np.uint8), (8, 8))
    with pytest.raises(ffi.BevwError):
        ex.warp()
