"""CPU-only soak of the JPEG kernels' lane code (tests/native/jpeg_emulate.cpp) on random files: every decode runs the three walkers of
bevw_jpeg_walk.h next to decode_sub on every subsequence and every entry state of the fixed point, the storing walker against decode_sub<true>,
and the result against Pillow's libjpeg-turbo.  No GPU.

    BEVW_SOAK_SECONDS=420 python tools/soak_jpeg_emulate.py [--seed 77]      -> profiles/r04/soak_jpeg_emulate.log"""
import argparse
import io
import os
import subprocess
import sys
import tempfile
import time

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import _native_build  # noqa: E402


def image(rng, h, w, kind):
    if kind == 0:
        return rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    if kind == 1:
        y, x = np.mgrid[0:h, 0:w]
        return np.stack([(x * 3 + y) % 256, (x + y * 2) % 256, (x * y // 7) % 256], -1).astype(np.uint8)
    if kind == 2:
        base = rng.integers(0, 256, (h // 16 + 1, w // 16 + 1, 3), dtype=np.uint8)
        return np.ascontiguousarray(np.kron(base, np.ones((16, 16, 1), np.uint8))[:h, :w])
    im = np.full((h, w, 3), rng.integers(0, 256), np.uint8)
    im[h // 3:, w // 4:] = rng.integers(0, 256, 3)
    return im


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=77)
    a = ap.parse_args()
    d = tempfile.mkdtemp(prefix="jpeg_emulate_soak_")
    exe, fin, fout = os.path.join(d, "jpeg_emulate"), os.path.join(d, "in.jpg"), os.path.join(d, "out.bin")
    _native_build.build(os.path.join(ROOT, "tests", "native", "jpeg_emulate.cpp"), exe)
    rng = np.random.default_rng(a.seed)
    t0 = time.time()
    t_end = t0 + float(os.environ.get("BEVW_SOAK_SECONDS", "300"))
    n = bad = walks = 0
    while time.time() < t_end:
        h, w = int(rng.integers(1, 300)), int(rng.integers(1, 300))
        sub, q = int(rng.integers(0, 3)), int(rng.choice([5, 30, 50, 75, 90, 95, 100]))
        kw = {}
        if rng.random() < 0.3:
            kw["restart_marker_blocks"] = int(rng.integers(1, 40))
        if rng.random() < 0.3:
            kw["optimize"] = True
        im = image(rng, h, w, int(rng.integers(0, 4)))
        b = io.BytesIO()
        try:
            Image.fromarray(im).save(b, "JPEG", quality=q, subsampling=sub, **kw)
        except OSError:      # Pillow's own output buffer is too small for some tiny images with extra markers: not a case
            continue
        raw = b.getvalue()
        open(fin, "wb").write(raw)
        r = subprocess.run([exe, "decode", fin, fout], capture_output=True, text=True)
        n += 1
        ok = r.returncode == 0 and "three walkers" in r.stdout
        if ok:
            buf = open(fout, "rb").read()
            ww, hh = np.frombuffer(buf[:8], np.int32)
            got = np.frombuffer(buf[16:], np.uint8).reshape(hh, ww, 3)
            ok = np.array_equal(got, np.asarray(Image.open(io.BytesIO(raw)).convert("RGB"))[:, :, ::-1])
            walks += int(r.stdout.split(" walks")[0].split()[-1])
        if not ok:
            bad += 1
            print("FAILED", dict(h=h, w=w, sub=sub, q=q, **kw), r.stderr[:300], flush=True)
    print(f"soak_jpeg_emulate: seed {a.seed}, {n} files, {walks} subsequence walks by decode_sub and the three walkers each, {bad} failures, {time.time() - t0:.0f} s")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
