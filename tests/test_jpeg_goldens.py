"""Row f4 against COMMITTED golden vectors (tests/golden/jpeg_goldens.npz, made by tests/golden/make_jpeg_goldens.py with Pillow's libjpeg-turbo):
the oracle on the CPU, the HIP codec on the GPU -- neither test imports Pillow."""
import hashlib
import os

import numpy as np
import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def G():
    return np.load(os.path.join(ROOT, "tests", "golden", "jpeg_goldens.npz"))


def _cases(G, prefix):
    return sorted(k[len(prefix):-4] for k in G.files if k.startswith(prefix) and k.endswith("_sha"))


def test_oracle_equals_the_goldens(G):
    from oracle import jpeg as JO

    dec, enc = _cases(G, "dec_"), _cases(G, "enc_")
    assert len(dec) >= 11 and len(enc) >= 5
    for name in dec:
        assert hashlib.sha256(JO.imdecode(G["dec_" + name].tobytes()).tobytes()).hexdigest() == str(G["dec_" + name + "_sha"]), name
    for name in enc:
        q, samp = (int(v) for v in G["enc_" + name + "_params"])
        assert hashlib.sha256(JO.imencode(G["enc_" + name], q, samp)).hexdigest() == str(G["enc_" + name + "_sha"]), name


@pytest.mark.gpu
def test_hip_codec_equals_the_goldens(G):
    from cameracalibration_amd import _ffi, imgcodecs

    _ffi.require_device()
    with imgcodecs.JpegCodec(0) as codec:
        for name in _cases(G, "dec_"):
            got = codec.decode([G["dec_" + name].tobytes()])[0]
            assert hashlib.sha256(got.tobytes()).hexdigest() == str(G["dec_" + name + "_sha"]), name
        for name in _cases(G, "enc_"):
            q, samp = (int(v) for v in G["enc_" + name + "_params"])
            f = codec.encode(G["enc_" + name][None], q, samp)[0]
            assert hashlib.sha256(f).hexdigest() == str(G["enc_" + name + "_sha"]), name
