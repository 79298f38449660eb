import hashlib
import io
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CAMS = ("front", "back", "left", "right")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "sanitize: the oracle / host emulators / JPEG marker parser under ASan + UBSan (CPU only)")


def decode_bgr(raw: bytes) -> np.ndarray:
    from PIL import Image

    rgb = np.asarray(Image.open(io.BytesIO(raw)).convert("RGB"))
    return np.ascontiguousarray(rgb[:, :, ::-1])


class RepoRig:
    """The reference's own sample rig (tests/golden/repo_rig.npz, built by tests/golden/make_fixtures.py)."""

    def __init__(self):
        z = np.load(os.path.join(ROOT, "tests", "golden", "repo_rig.npz"))
        self._z = z
        self.rig = {n: (z[f"{n}_K"], z[f"{n}_D"], z[f"{n}_H"]) for n in CAMS}
        self.expected_sha = dict((l.split()[0], l.split()[2]) for l in z["decoded_sha256"].tolist())
        self._cache = {}

    def image(self, key: str) -> np.ndarray:
        if key not in self._cache:
            arr = decode_bgr(self._z[f"{key}_img"].tobytes())
            got = hashlib.sha256(arr.tobytes()).hexdigest()
            assert got == self.expected_sha[f"{key}_img"], f"decoder mismatch for {key}: fixture is not valid here"
            self._cache[key] = arr
        return self._cache[key]

    def frames(self):
        return [self.image(n) for n in CAMS]


@pytest.fixture(scope="session")
def repo_rig():
    return RepoRig()


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O

    O.build()
    return O
