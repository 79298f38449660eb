#!/usr/bin/env python3
"""Random-case soak of the GPU JPEG codec (row f4) against Pillow's libjpeg-turbo, on the GPU box:

    python tools/soak_jpeg.py [--cases 2000] [--seed 1]

Every case draws a size (1 .. 700 per side, now and then camera-sized), a content kind, a quality, a sampling, optionally a restart interval and
private (optimised) Huffman tables, and a batch size; the files are written by Pillow, decoded by the engine in one batch and compared byte for
byte with Pillow's decode; the images are encoded by the engine and compared byte for byte with the files Pillow writes.  Prints one summary line."""
import argparse
import io
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from PIL import Image  # noqa: E402

from cameracalibration_amd import imgcodecs  # noqa: E402


def image(rng, h, w, kind):
    y, x = np.mgrid[0:h, 0:w]
    if kind == 0:
        a = rng.integers(0, 256, (h, w, 3))
    elif kind == 1:
        f = rng.uniform(2, 40, 3)
        a = 128 + 110 * np.sin(x[..., None] / f) * np.cos(y[..., None] / f[::-1]) + rng.normal(0, rng.uniform(0, 12), (h, w, 3))
    elif kind == 2:
        a = np.full((h, w, 3), rng.integers(0, 256, 3))       # flat: blocks of 4 - 6 bits, thousands of them per subsequence
    else:
        a = (x[..., None] * rng.integers(1, 5, 3) + y[..., None] * rng.integers(1, 5, 3)) % 256
        a[rng.integers(0, h):, :] = 255 * (rng.integers(0, 2, 3))   # saturated areas: 0xFF bytes in the stream
    return np.clip(a, 0, 255).astype(np.uint8)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=2000)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    codec = imgcodecs.JpegCodec(0)
    bad = files_total = skipped = 0
    t0 = time.time()
    done = 0
    t_end = t0 + float(os.environ.get("BEVW_SOAK_SECONDS", "1e9"))   # stop there and report what was done
    for case in range(a.cases):
        if time.time() > t_end:
            break
        done += 1
        if rng.random() < 0.03:
            h, w = int(rng.choice([960, 1024, 1080])), int(rng.choice([1280, 1080]))
        else:
            h, w = int(rng.integers(1, 700)), int(rng.integers(1, 700))
        sub, samp = [(2, 0x22), (1, 0x21), (0, 0x11)][int(rng.integers(0, 3))]
        q = int(rng.choice([5, 30, 50, 75, 90, 95, 100]))
        n = int(rng.integers(1, 6))
        kw = {}
        if rng.random() < 0.25:
            kw["restart_marker_blocks"] = int(rng.integers(1, 40))
        if rng.random() < 0.25:
            kw["optimize"] = True
        ims = [image(rng, h, w, int(rng.integers(0, 4))) for _ in range(n)]
        files = []
        try:
            for im in ims:
                b = io.BytesIO()
                Image.fromarray(np.ascontiguousarray(im[:, :, ::-1])).save(b, "JPEG", quality=q, subsampling=sub, **kw)
                files.append(b.getvalue())
        except OSError:      # Pillow's own output buffer is too small for some tiny images with extra markers: not a case
            skipped += 1
            continue
        try:
            got = codec.decode(files)
        except Exception as e:      # a refusal of a valid file is a failure of the case: report it, keep the files, go on
            bad += 1
            print("FAILED case", case, dict(h=h, w=w, sub=sub, q=q, n=n, **kw), repr(e), flush=True)
            out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "soak_jpeg_fail_seed%d_case%d.npz" % (a.seed, case))
            if bad <= 3:
                os.makedirs(os.path.dirname(out), exist_ok=True)
                np.savez(out, **{"f%d" % i: np.frombuffer(f, np.uint8) for i, f in enumerate(files)})
            continue
        ok = all(np.array_equal(got[i], np.asarray(Image.open(io.BytesIO(f)).convert("RGB"))[:, :, ::-1]) for i, f in enumerate(files))
        if "optimize" not in kw and "restart_marker_blocks" not in kw:      # the engine writes libjpeg's default file: standard tables, no DRI
            enc = codec.encode(np.stack(ims), q, samp)
            ok = ok and all(enc[i] == files[i] for i in range(n))
        files_total += n
        if not ok:
            bad += 1
            dec_bad = [i for i, f in enumerate(files) if not np.array_equal(got[i], np.asarray(Image.open(io.BytesIO(f)).convert("RGB"))[:, :, ::-1])]
            print("MISMATCH case", case, dict(h=h, w=w, sub=sub, q=q, n=n, **kw), "decode differs for files", dec_bad, flush=True)
            if bad <= 3:
                out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "soak_jpeg_fail_seed%d_case%d.npz" % (a.seed, case))
                os.makedirs(os.path.dirname(out), exist_ok=True)
                np.savez(out, **{"f%d" % i: np.frombuffer(f, np.uint8) for i, f in enumerate(files)})
                again = codec.decode(files)      # (the same files a second time: is the result reproducible?)
                print("   second decode of the same files: differs for", [i for i, f in enumerate(files) if not np.array_equal(again[i], np.asarray(Image.open(io.BytesIO(f)).convert("RGB"))[:, :, ::-1])], flush=True)
    print(f"soak_jpeg: seed {a.seed}, {done} of {a.cases} cases run, {files_total} files ({skipped} cases Pillow could not write), {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
