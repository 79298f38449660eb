"""The unit kernels' integer blend weight checked WITHOUT a GPU, exhaustively (round 6).

BlendMask.__call__ (surroundBEV.py:187-188, 279-280) is trunc(f32(v) * f32(m / 255.0)); the unit kernels compute (v * (m * 32897)) >> 23
(blend_apply_q23, cameracalibration_amd/csrc/bevw_device.h).  tests/native/blend_exhaustive.cpp compiles the header's functions for the
host and compares all 65,536 (v, m) pairs; the NumPy statement of the reference's expression is compared here as well."""
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def test_integer_blend_identity_numpy():
    v = np.arange(256, dtype=np.uint8)[:, None]
    m = np.arange(256, dtype=np.uint8)[None, :]
    weight = (m / 255.0).astype(np.float32)                      # BlendMask: self.weight = np.float32(mask / 255.0) per channel
    ref = (v * weight).astype(np.uint8).astype(np.int64)         # (img * self.weight).astype(np.uint8)
    vi, mi = v.astype(np.int64), m.astype(np.int64)
    assert np.array_equal(ref, (vi * mi) // 255)
    assert np.array_equal(ref, (vi * (mi * 32897)) >> 23)
    assert int((mi * 32897).max()) < 1 << 24 and int((vi * (mi * 32897)).max()) < 1 << 32


@pytest.mark.skipif(not (os.path.exists(HIPCC) or shutil.which("hipcc")), reason="hipcc not available")
def test_integer_blend_identity_device_header(tmp_path):
    from tests import _native_build

    exe = str(tmp_path / "blend_exhaustive")
    _native_build.build(os.path.join(ROOT, "tests", "native", "blend_exhaustive.cpp"), exe)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "blend q23 ok: 65536 pairs" in r.stdout, r.stdout
