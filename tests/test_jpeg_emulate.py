"""CPU run of the JPEG kernels' per-lane code (tests/native/jpeg_emulate.cpp): the functions in cameracalibration_amd/csrc/bevw_jpeg.h
and bevw_jpeg_walk.h are __host__ __device__, so the parallel Huffman decoder's fixed point, the inverse / forward DCT, the colour code and
the bit writer the GPU runs are checked here, without a GPU, against Pillow's libjpeg-turbo (row f4).  Every decode below also runs the
three walkers the kernels use (straight-line lane walker, scalar walker, storing walker) next to decode_sub, the plain statement of the
algorithm, on every subsequence and every entry state of the fixed point, and fails on the first difference."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from tests import _jpeg_common as JC

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
pytestmark = pytest.mark.skipif(not (os.path.exists(HIPCC) or shutil.which("hipcc")), reason="hipcc not available")
pytest.importorskip("PIL")


@pytest.fixture(scope="module")
def emu(tmp_path_factory):
    d = tmp_path_factory.mktemp("jpeg_emulate")
    exe = str(d / "jpeg_emulate")
    from tests import _native_build

    _native_build.build(os.path.join(ROOT, "tests", "native", "jpeg_emulate.cpp"), exe)

    class Emu:
        def decode(self, raw):
            a, b = str(d / "in.jpg"), str(d / "out.bin")
            open(a, "wb").write(raw)
            r = subprocess.run([exe, "decode", a, b], capture_output=True, text=True, timeout=300)
            assert r.returncode == 0, r.stdout + r.stderr
            assert "walks by three walkers each" in r.stdout, r.stdout
            buf = open(b, "rb").read()
            w, h, rounds, nsub = np.frombuffer(buf[:16], np.int32)
            return np.frombuffer(buf[16:], np.uint8).reshape(h, w, 3), int(rounds), int(nsub)

        def unstuff(self, data: bytes):
            a = str(d / "raw.bin")
            open(a, "wb").write(data)
            r = subprocess.run([exe, "unstuff", a], capture_output=True, text=True, timeout=120)
            assert r.returncode == 0, r.stdout + r.stderr
            return r.stdout

        def encode(self, im, q=95, samp=0x22):
            a, b = str(d / "in.bin"), str(d / "out.jpg")
            h, w = im.shape[:2]
            open(a, "wb").write(np.array([w, h, q, samp], np.int32).tobytes() + np.ascontiguousarray(im).tobytes())
            r = subprocess.run([exe, "encode", a, b], capture_output=True, text=True, timeout=300)
            assert r.returncode == 0, r.stdout + r.stderr
            return open(b, "rb").read()

    return Emu()


def test_parallel_huffman_decoder_reaches_the_sequential_decoders_states(emu):
    for name, raw in JC.repo_camera_jpegs().items():
        got, rounds, nsub = emu.decode(raw)
        assert np.array_equal(got, JC.pil_decode(raw)), name
        assert nsub > 500 and 1 <= rounds <= nsub   # ~1 KB of entropy-coded data per lane; a handful of rounds, not nsub of them


@pytest.mark.parametrize("sub,samp", JC.SUBSAMPLINGS)
def test_emulated_kernels_equal_libjpeg_turbo(emu, sub, samp):
    for h, w in ((8, 8), (1, 1), (3, 5), (17, 33), (40, 56), (100, 75), (64, 2), (4, 5), (120, 200)):
        for kind in (0, 1, 2):
            q = (95, 30, 100)[kind]
            im = JC.image(h, w, kind)
            f = JC.pil_encode(im, q, sub)
            assert np.array_equal(emu.decode(f)[0], JC.pil_decode(f)), (h, w, kind)
            assert emu.encode(im, q, samp) == f, (h, w, kind)


def test_emulated_restart_segments_and_grey(emu):
    im = JC.image(200, 300, 2)
    f = JC.pil_encode_gray(im[:, :, 0])
    assert np.array_equal(emu.decode(f)[0], JC.pil_decode(f))
    for kw in (dict(restart_marker_blocks=1), dict(restart_marker_blocks=3), dict(restart_marker_rows=1)):
        f = JC.pil_encode(im, 90, 2, **kw)
        assert np.array_equal(emu.decode(f)[0], JC.pil_decode(f)), kw


def test_emulated_private_huffman_tables_and_long_codes(emu):
    # optimize=True: file-specific tables (other code-length profiles than Annex K); quality 100 noise: 16-bit codes in every block
    for k, q in ((0, 85), (1, 100), (2, 40), (1, 97)):
        im = JC.image(136, 200, k)
        f = JC.pil_encode(im, q, 2, optimize=True)
        assert np.array_equal(emu.decode(f)[0], JC.pil_decode(f)), (k, q)
    f = JC.pil_encode(JC.image(96, 160, 1), 100, 0)
    assert np.array_equal(emu.decode(f)[0], JC.pil_decode(f))
    # uniform noise at quality 100: blocks longer than a 1024-bit subsequence (lanes that own no block; owners walking through their successor)
    rng = np.random.default_rng(20260925)
    for h, w, sub in ((84, 542, 2), (120, 96, 0), (64, 200, 1)):
        f = JC.pil_encode(rng.integers(0, 256, (h, w, 3), dtype=np.uint8), 100, sub)
        got, rounds, nsub = emu.decode(f)
        assert np.array_equal(got, JC.pil_decode(f)), (h, w, sub)
        assert rounds > 8


def test_emulated_unstuffing_on_byte_soup(emu):
    # the un-stuffing kernels' lane code against the sequential statement on bytes rich in 0xFF, stuffed zeros, RSTn, fill bytes and early
    # terminators -- more than any encoder would write, and placed on every position relative to the 16-byte lanes
    rng = np.random.default_rng(3)
    alphabet = np.array([0xFF] * 6 + [0x00] * 4 + [0xD0, 0xD3, 0xD7, 0xD9, 0xC4, 0x01, 0x7F, 0x80], np.uint8)
    for trial in range(60):
        n = int(rng.integers(1, 400))
        soup = alphabet[rng.integers(0, len(alphabet), n)] if trial % 2 else rng.integers(0, 256, n, dtype=np.uint8)
        emu.unstuff(bytes(soup))
    # a stream as an encoder writes it: no terminator inside, a marker at the very end
    body = bytes(b for x in rng.integers(0, 256, 5000, dtype=np.uint8) for b in ((255, 0) if x == 255 else (int(x),)))
    assert "1 segments" in emu.unstuff(body + b"\xff\xd9")
    assert "3 segments" in emu.unstuff(body[:1000] + b"\xff\xd0" + body[1000:3001] + b"\xff\xff\xd1" + body[3001:] + b"\xff\xd9")


def test_emulated_bev_sized_file(emu):
    im = JC.image(1080, 1080, 2)
    assert emu.encode(im) == JC.pil_encode(im)
