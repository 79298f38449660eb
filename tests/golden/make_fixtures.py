#!/usr/bin/env python3
"""Build tests/golden/repo_rig.npz from the reference's own sample data.

Run in the build container only (it reads /root/reference, which does not exist on the GPU box):

    python tests/golden/make_fixtures.py

What it stores (inputs only -- the reference publishes no golden OUTPUTS):
  * K/D/H float64 calibration of the four sample cameras
    (reference: SurroundBirdEyeView/data/<cam>/camera_<cam>_{K,D,H}.npy, loaded at surroundBEV.py:83-85)
  * the four 1280x1024 camera frames and the car sprite as their ORIGINAL encoded bytes
    (decoded at test time with Pillow/libjpeg-turbo, the decoder cv2.imread wraps; main.py:74-77)
  * one intrinsic-calibration frame and the extrinsic src/dst pair (main.py:23,53-54)
  * sha256 of every decoded BGR array so a different decoder build is detected, not silently accepted.
"""
import hashlib
import io
import os
import sys

import numpy as np
from PIL import Image

REF = os.environ.get("BEVW_REFERENCE_ROOT", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "repo_rig.npz")
CAMS = ("front", "back", "left", "right")


def decode_bgr(raw: bytes) -> np.ndarray:
    """cv2.imread(path) equivalent: 8-bit, 3 channels, BGR order, alpha dropped."""
    rgb = np.asarray(Image.open(io.BytesIO(raw)).convert("RGB"))
    return np.ascontiguousarray(rgb[:, :, ::-1])


def main() -> int:
    if not os.path.isdir(REF):
        print("reference tree not found at", REF, file=sys.stderr)
        return 1
    blob = {}
    base = os.path.join(REF, "SurroundBirdEyeView", "data")
    for cam in CAMS:
        for kind in "KDH":
            blob[f"{cam}_{kind}"] = np.load(os.path.join(base, cam, f"camera_{cam}_{kind}.npy"))
        with open(os.path.join(base, cam, f"{cam}.jpg"), "rb") as fh:
            blob[f"{cam}_img"] = np.frombuffer(fh.read(), dtype=np.uint8)
    extra = {
        "car_img": os.path.join(base, "car.jpg"),
        "incalib_img": os.path.join(REF, "IntrinsicCalibration", "data", "img_raw0.jpg"),
        "excalib_src_img": os.path.join(REF, "ExtrinsicCalibration", "data", "img_src_back.jpg"),
    }
    for key, path in extra.items():
        with open(path, "rb") as fh:
            blob[key] = np.frombuffer(fh.read(), dtype=np.uint8)
    digests = []
    for key in sorted(k for k in blob if k.endswith("_img")):
        arr = decode_bgr(blob[key].tobytes())
        digests.append(f"{key} {arr.shape[0]}x{arr.shape[1]} {hashlib.sha256(arr.tobytes()).hexdigest()}")
    blob["decoded_sha256"] = np.array(digests)
    np.savez(OUT, **blob)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")
    for d in digests:
        print(" ", d)
    return 0


if __name__ == "__main__":
    sys.exit(main())
