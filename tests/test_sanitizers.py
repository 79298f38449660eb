"""Sanitizer leg of the CPU suite: the C oracle, the host emulators of the kernels' lane code and the host JPEG
marker parser under AddressSanitizer + UndefinedBehaviorSanitizer.  Any finding aborts the sanitized process and fails the test.

  * oracle/*.c            gcc -fsanitize=address,undefined (oracle.sanitized_build), loaded into a python child with libasan preloaded;
                          the known-answer, variant and libjpeg-turbo pin tests run on it
  * tests/native/*.cpp    clang (hipcc --cuda-host-only) -fsanitize=address,undefined: the JPEG emulator tests + a short soak of random
                          files, the unit-schedule emulator on its synthetic rigs and a fuzz of random small rigs
  * the marker parser     tests/native/jpeg_parse_fuzz.cpp: mutated files (byte flips, truncations, segment lengths, DHT code counts,
                          stray markers) through parse_header / make_hufftab / unstuff_scan, the host code that reads untrusted bytes
                          (cameracalibration_amd/csrc/bevw_jpeg.h).  Round 5: it found make_hufftab writing beyond HuffTab::fast for
                          over-subscribed code counts (fixed; the case stays in the mutation set)."""
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
pytestmark = [pytest.mark.sanitize]
ASAN_OPTIONS = "detect_leaks=0:abort_on_error=0:halt_on_error=1"


def _gcc_runtime(name):
    r = subprocess.run(["gcc", "-print-file-name=" + name], capture_output=True, text=True)
    p = r.stdout.strip()
    return p if r.returncode == 0 and os.path.isabs(p) and os.path.exists(p) else None


def _child_env(**extra):
    e = dict(os.environ)
    e.update(ASAN_OPTIONS=ASAN_OPTIONS, UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1", PYTHONPATH=ROOT)
    e.update(extra)
    return e


@pytest.mark.skipif(shutil.which("gcc") is None or _gcc_runtime("libasan.so") is None, reason="gcc / libasan not available")
def test_oracle_under_asan_ubsan():
    """The oracle's own test modules, with oracle/_san/lib*_san.so instead of the parity builds."""
    env = _child_env(BEVW_ORACLE_SANITIZE="1", LD_PRELOAD=_gcc_runtime("libasan.so"), OMP_NUM_THREADS="4")
    probe = subprocess.run([sys.executable, "-c", "from oracle import oracle as O, jpeg as J; print(O.build()); print(J.build()); O.lib(); J.lib()"],
                           cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert probe.returncode == 0 and "_san.so" in probe.stdout, probe.stdout + probe.stderr
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-p", "no:cacheprovider", "tests/test_oracle_known_answers.py",
                        "tests/test_oracle_variants.py", "tests/test_jpeg_oracle.py"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    assert " passed" in r.stdout and "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, (r.stdout + r.stderr)[-4000:]


needs_hipcc = pytest.mark.skipif(not (os.path.exists(HIPCC) or shutil.which("hipcc")), reason="hipcc not available")


@needs_hipcc
def test_emulators_under_asan_ubsan(tmp_path):
    """tests/native/jpeg_emulate.cpp and unit_emulate.cpp rebuilt with the sanitizers (BEVW_NATIVE_SANITIZE=1 makes tests/_native_build.py
    do so): their own test modules, then random small rigs through the unit compiler + emulator and a bounded soak of random JPEG files."""
    env = _child_env(BEVW_NATIVE_SANITIZE="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-p", "no:cacheprovider", "tests/test_jpeg_emulate.py",
                        "tests/test_unit_schedule.py::test_units_on_synthetic_tables"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0 and " passed" in r.stdout, (r.stdout + r.stderr)[-4000:]
    from tests import _native_build

    exe = str(tmp_path / "unit_emulate_san")
    _native_build.build(os.path.join(ROOT, "tests", "native", "unit_emulate.cpp"), exe, sanitize=True)
    f = subprocess.run([exe], env=_child_env(BEVW_EMU_FUZZ="5 24"), capture_output=True, text=True, timeout=900)
    assert f.returncode == 0 and f.stdout.count("unit schedule ok") == 24, (f.stdout + f.stderr)[-3000:]
    s = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "soak_jpeg_emulate.py"), "--seed", "91"], cwd=ROOT,
                       env=_child_env(BEVW_NATIVE_SANITIZE="1", BEVW_SOAK_SECONDS="45"), capture_output=True, text=True, timeout=900)
    assert s.returncode == 0 and "0 failures" in s.stdout, (s.stdout + s.stderr)[-3000:]


@needs_hipcc
def test_jpeg_marker_parser_fuzz_under_asan_ubsan(tmp_path):
    pytest.importorskip("PIL")
    from PIL import Image
    from tests import _native_build

    exe = str(tmp_path / "jpeg_parse_fuzz")
    _native_build.build(os.path.join(ROOT, "tests", "native", "jpeg_parse_fuzz.cpp"), exe, sanitize=True)
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (96, 136, 3), dtype=np.uint8)
    files = []
    ex = Image.Exif()
    ex[0x0112] = 6
    for i, kw in enumerate([dict(quality=90, subsampling=2), dict(quality=75, subsampling=0, restart_marker_blocks=3), dict(quality=85, subsampling=1, exif=ex),
                            dict(quality=30, subsampling=2, optimize=True), dict(quality=95, subsampling=2, restart_marker_rows=1)]):
        p = str(tmp_path / ("seed%d.jpg" % i))
        Image.fromarray(img).save(p, **kw)
        files.append(p)
    p = str(tmp_path / "grey.jpg")
    Image.fromarray(img[:, :, 0]).save(p, quality=60)
    files.append(p)
    from conftest import RepoRig
    z = RepoRig()._z
    for n in ("front", "right"):   # two of the reference's own camera files
        p = str(tmp_path / (n + ".jpg"))
        open(p, "wb").write(z[n + "_img"].tobytes())
        files.append(p)
    r = subprocess.run([exe, "7", "6000"] + files, env=_child_env(), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "jpeg parser fuzz ok" in r.stdout, (r.stdout + r.stderr)[-3000:]
