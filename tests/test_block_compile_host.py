"""Host-side plan compiler of the block tiles (cameracalibration_amd/csrc/bevw_block.h: block_compile) checked WITHOUT a GPU:
tests/native/block_compile_check.cpp is compiled with hipcc (the host part is all that runs) and verifies, on synthetic LUTs with a
closed-form answer, that claimed base tiles, group lists and every pixel's two LDS pair addresses are what the kernel expects."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.mark.skipif(not (os.path.exists(HIPCC) or shutil.which("hipcc")), reason="hipcc not available")
def test_block_compile_on_synthetic_tables(tmp_path):
    exe = str(tmp_path / "block_compile_check")
    from tests import _native_build

    _native_build.build(os.path.join(ROOT, "tests", "native", "block_compile_check.cpp"), exe)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "block_compile ok" in r.stdout
