#!/usr/bin/env python3
"""Pin the oracle against a REAL OpenCV: run wherever `import cv2` works and a checkout of the reference exists.

    BEVW_REFERENCE_ROOT=/path/to/CameraCalibration python tests/golden/make_goldens_with_cv2.py

Writes tests/golden/cv2_goldens.npz (digests + small probes, a few hundred KB) from the reference's OWN modules
(SurroundBirdEyeView/surroundBEV.py, ExtrinsicCalibration/extrinsicCalib.py) and direct cv2 calls, on the inputs of
tests/golden/repo_rig.npz.  tests/test_cv2_goldens.py then compares the CPU oracle with it (and skips while the file is
absent -- this image has no cv2, so parity with OpenCV itself is still unpinned: DESIGN.md section 2).
Commit the .npz together with the OpenCV version it prints.
"""
import importlib.util
import io
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _golden_cases as GC  # noqa: E402

REF = os.environ.get("BEVW_REFERENCE_ROOT", "/root/reference")
OUT = os.environ.get("BEVW_GOLDENS_OUT", os.path.join(HERE, "cv2_goldens.npz"))


def decode_bgr(raw: bytes) -> np.ndarray:
    from PIL import Image

    rgb = np.asarray(Image.open(io.BytesIO(raw)).convert("RGB"))
    return np.ascontiguousarray(rgb[:, :, ::-1])


def load_module(relpath, name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, relpath))
    mod = importlib.util.module_from_spec(spec)
    argv, sys.argv = sys.argv, [sys.argv[0]]   # the reference parses its command line at import time
    try:
        spec.loader.exec_module(mod)
    finally:
        sys.argv = argv
    return mod


def main() -> int:
    import cv2

    if getattr(cv2, "IS_ORACLE_SHIM", False) and os.environ.get("BEVW_GOLDENS_ALLOW_SHIM") != "1":
        print("refusing to write goldens from the oracle-built cv2 stand-in", file=sys.stderr)
        return 2
    z = np.load(os.path.join(HERE, "repo_rig.npz"))
    img = {k[:-4]: decode_bgr(z[k].tobytes()) for k in z.files if k.endswith("_img")}
    frames = [img[n] for n in GC.CAMS]
    sb = load_module("SurroundBirdEyeView/surroundBEV.py", "reference_surroundBEV")
    ec = load_module("ExtrinsicCalibration/extrinsicCalib.py", "reference_extrinsicCalib")
    cases = {}

    # tables, masks and single-camera remaps of the sample rig (module defaults: 1280x1024 -> 1000x1000, car 250x400)
    args = sb.BevGenerator.get_args()
    args.CAR_WIDTH, args.CAR_HEIGHT = 250, 400
    for blend in (False, True):
        gen = sb.BevGenerator(blend=blend, balance=False)
        for i, n in enumerate(GC.CAMS):
            cases["mask_%s_%s" % ("blend" if blend else "direct", n)] = gen.masks[i].mask
            if not blend:
                cam = gen.cameras[i]
                cases["und_map1_" + n], cases["und_map2_" + n] = cam.undistort_maps
                cases["bev_map1_" + n], cases["bev_map2_" + n] = cam.bev_maps
                cases["raw2bev_" + n] = cam.raw2bev(frames[i])
        if not blend:
            cases["undistort_front"] = gen.cameras[0].undistort(frames[0])

    # BevGenerator.__call__ in all four modes, with main.py's car size, with and without the car sprite
    args.CAR_WIDTH, args.CAR_HEIGHT = GC.MAIN_CAR
    car = sb.padding(img["car"], args.BEV_WIDTH, args.BEV_HEIGHT)
    cases["car_padded"] = car
    for blend, balance in GC.MODES:
        gen = sb.BevGenerator(blend=blend, balance=balance)
        tag = "bev_%d%d" % (blend, balance)
        cases[tag] = gen(*[f.copy() for f in frames])
        cases[tag + "_car"] = gen(*[f.copy() for f in frames], car=car)
    for n, out in zip(GC.CAMS, sb.luminance_balance([f.copy() for f in frames])):
        cases["lum_" + n] = out
    cases["color_balance_back"] = sb.color_balance(img["back"].copy())
    ramp = np.arange(256, dtype=np.uint8).reshape(256, 1)
    for k, g in enumerate(GC.ADDWEIGHTED_GAINS):      # the call shape of surroundBEV.py:52-54, scalar src2
        cases["addweighted_scalar_%d" % k] = cv2.addWeighted(ramp, g, 0, 0, 0)

    # InCalibrator.undistort geometry (intrinsicCalib.py:90-103 with FOCAL_SCALE 0.5, SIZE_SCALE 1) and the pinhole maps
    K, D = z["front_K"], z["front_D"]
    src = img["incalib"]
    h, w = src.shape[:2]
    Kd = K.copy()
    Kd[0, 0] *= 0.5
    Kd[1, 1] *= 0.5
    Kd[0, 2], Kd[1, 2] = w / 2, h / 2
    m1, m2 = cv2.fisheye.initUndistortRectifyMap(K, D, np.eye(3), Kd, (w, h), cv2.CV_16SC2)
    cases["incalib_undistort"] = cv2.remap(src, m1, m2, interpolation=cv2.INTER_LINEAR)
    p1, p2 = cv2.initUndistortRectifyMap(K, np.array(GC.PINHOLE_D), np.eye(3), Kd, (w, h), cv2.CV_16SC2)
    cases["pinhole_map1"], cases["pinhole_map2"] = p1, p2

    # ExCalibrator.warp and the pre-processing warps
    cases["excalib_warp"] = cv2.warpPerspective(img["excalib_src"], z["back_H"], (1000, 1000))
    small = np.ascontiguousarray(img["back"][:301, :403])
    center = ec.CenterImage.__new__(ec.CenterImage)
    center.x, center.y = 100, 250
    cases["translate"] = center.translate(small)
    for f in GC.RESIZE_FACTORS:
        cases["resize_%g" % f] = cv2.resize(small, (0, 0), fx=f, fy=f)

    # row f4: cv2's JPEG codec itself (main.py:74-77 cv2.imread of the camera files; surroundBEV.py:340 cv2.imwrite of the stitched image).  The
    # JPEG oracle is pinned against Pillow's libjpeg-turbo; these cases say whether THIS OpenCV's bundled codec decodes / writes the same bytes.
    for n in GC.CAMS:
        cases["imread_" + n] = cv2.imdecode(np.frombuffer(z[n + "_img"].tobytes(), np.uint8), cv2.IMREAD_COLOR)
    ok, enc = cv2.imencode(".jpg", GC.jpeg_test_image())                       # cv2.imwrite's defaults: quality 95, 4:2:0
    cases["imwrite_default"] = np.ascontiguousarray(enc).reshape(-1)
    ok, enc = cv2.imencode(".jpg", GC.jpeg_test_image(), [cv2.IMWRITE_JPEG_QUALITY, 100])   # Tools/undistort.py:73 with its default -quality 100
    cases["imwrite_q100"] = np.ascontiguousarray(enc).reshape(-1)

    # implementation probes (OpenCV >= 4.11 rewrote warpPerspective and reworked remap): small cases stored whole, see _golden_cases.py
    u16, s16, u8 = GC.probe_images()
    Hp = np.array(GC.PROBE_H)
    full = {"probe_warp_16uc1": cv2.warpPerspective(u16, Hp, GC.PROBE_DSIZE),      # surroundBEV.py:105-108: the map2 half of the BEV LUT
            "probe_warp_16sc2": cv2.warpPerspective(s16, Hp, GC.PROBE_DSIZE),      #                          the map1 half
            "probe_warp_8uc3": cv2.warpPerspective(u8, Hp, GC.PROBE_DSIZE)}        # extrinsicCalib.py:166-169
    pm1, pm2 = GC.probe_maps()
    full["probe_remap_8uc3"] = cv2.remap(u8, pm1, pm2, interpolation=cv2.INTER_LINEAR)   # surroundBEV.py:113-117

    blob = GC.pack(cases)
    blob.update(GC.pack_full(full))
    blob["cv2_version"] = np.array(getattr(cv2, "__version__", "oracle-shim"))
    np.savez_compressed(OUT, **blob)
    print("wrote", OUT, os.path.getsize(OUT), "bytes,", len(cases), "cases, OpenCV", blob["cv2_version"])
    return 0


if __name__ == "__main__":
    sys.exit(main())
