"""INDEPENDENT cross-checks of the CPU oracle ("pin the oracle or shrink what is unpinned", item 2c).

cv2 cannot be installed in this image, so parity with a real OpenCV stays unpinned until tests/golden/cv2_goldens.npz
arrives (tests/golden/README.md).  What CAN be done here is to hold every primitive of the oracle against an
implementation written by OTHER people with a different algorithm -- none of the code below shares a line with
oracle/bevoracle.c or oracle/np_twin.py:

  remap (A.4)              scipy.ndimage.map_coordinates(order=1) in float64 at the LUT's own sub-pixel positions.  The
                           5-bit x 5-bit weights are exact in float64, so away from the border this is an EXACT check of the
                           fixed-point arithmetic (floor(v + 1/2) of the true bilinear value), not a closeness check.
  warpPerspective (A.2)    the same interpolator at the EXACT fp64 H^-1 coordinates: differs only by OpenCV's 1/32-pixel
                           coordinate quantisation; reported as mean / max |diff|.
  fillPoly (A.5)           PIL.ImageDraw.polygon (a different scan converter): interiors identical, every difference on the
                           polygon boundary; the boundary differences are listed, not hidden.
  fisheye maps (A.1)       inverse consistency: the map's target point is pushed back through an independent numerical
                           inverse of the distortion polynomial (scipy.optimize.brentq) and must land on the pixel it came from.
  BGR->HSV (A.7)           skimage.color.rgb2hsv from the image's conda python 3.9 (when present): |dH| <= 1, |dS| <= 1, V exact.
  HSV->BGR (A.7)           skimage.color.hsv2rgb in float64: the 8-bit result is its rounding (<= 0.5 LSB).
  resize (A.10)            skimage.transform.resize(order=1) where both sampling conventions coincide, scipy at cv2's own positions elsewhere.

These are evidence ABOUT the restatement, reported as measured differences; they are not the parity oracle.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
from scipy import ndimage, optimize

from cameracalibration_amd import workloads as W


def _float_positions(map1, map2):
    code = map2.astype(np.int64) & 1023
    return map1[..., 0].astype(np.float64) + (code & 31) / 32.0, map1[..., 1].astype(np.float64) + (code >> 5) / 32.0


def _bilinear(img, xs, ys):
    out = np.empty(xs.shape + (3,), np.float64)
    for c in range(3):
        out[..., c] = ndimage.map_coordinates(img[..., c].astype(np.float64), [ys, xs], order=1, mode="grid-constant", cval=0.0)
    return out


@pytest.mark.parametrize("cam", ["front", "right"])
def test_remap_is_exactly_the_rounded_float_bilinear(oracle, repo_rig, cam):
    """cv2.remap(img, bev_map1, bev_map2, INTER_LINEAR) at surroundBEV.py:117 on the reference's own frame and LUT."""
    K, D, H = repo_rig.rig[cam]
    ref = oracle.RefCamera(K, D, H, dict(oracle.DEFAULT_CFG))
    img = repo_rig.image(cam)
    m1, m2 = ref.bev_maps
    got = oracle.remap(img, m1, m2)
    xs, ys = _float_positions(m1, m2)
    want = np.floor(_bilinear(img, xs, ys) + 0.5)
    h, w = img.shape[:2]
    interior = (m1[..., 0] >= 0) & (m1[..., 0] < w - 1) & (m1[..., 1] >= 0) & (m1[..., 1] < h - 1)
    assert interior.mean() > 0.5
    d = np.abs(got.astype(np.float64) - want)[interior]
    assert d.max() == 0, f"{int(np.count_nonzero(d))} interior samples differ, max {d.max()}"
    # border taps (BORDER_CONSTANT 0 per tap): the float interpolator with zero padding agrees there as well
    dborder = np.abs(got.astype(np.float64) - want)[~interior]
    assert dborder.size == 0 or dborder.max() <= 1


def test_warp_perspective_close_to_exact_coordinates(oracle, repo_rig):
    """ExCalibrator.warp (extrinsicCalib.py:166-169): the oracle quantises source coordinates to 1/32 pixel as OpenCV does; an
    interpolator fed the exact H^-1 coordinates may therefore differ by |gradient| / 64 -- small on average, a few LSB on edges."""
    src = repo_rig.image("excalib_src")
    H = repo_rig.rig["back"][2]
    got = oracle.warp_perspective(src, H, (1000, 1000)).astype(np.float64)
    Hi = np.linalg.inv(H)
    yy, xx = np.mgrid[0:1000, 0:1000].astype(np.float64)
    den = Hi[2, 0] * xx + Hi[2, 1] * yy + Hi[2, 2]
    xs = (Hi[0, 0] * xx + Hi[0, 1] * yy + Hi[0, 2]) / den
    ys = (Hi[1, 0] * xx + Hi[1, 1] * yy + Hi[1, 2]) / den
    want = _bilinear(src, xs, ys)
    h, w = src.shape[:2]
    inside = (xs >= 1) & (xs < w - 2) & (ys >= 1) & (ys < h - 2)
    d = np.abs(got - want)[inside]
    print("warpPerspective vs exact-coordinate bilinear: mean |diff| %.3f, 99.9 %% <= %.1f, max %.1f LSB over %d samples"
          % (d.mean(), np.percentile(d, 99.9), d.max(), d.size))
    assert inside.mean() > 0.5
    assert d.mean() < 0.6            # rounding (0.25 on average) + 1/64-pixel coordinate error
    assert np.percentile(d, 99.9) <= 6.0


@pytest.mark.parametrize("geo", [(1000, 1000, 250, 400), (1080, 1080, 270, 432), (640, 480, 100, 160)])
@pytest.mark.parametrize("blend", [False, True])
def test_fill_poly_against_pil(oracle, geo, blend):
    from PIL import Image, ImageDraw

    bw, bh = geo[0], geo[1]
    worst = 0
    for name in oracle.CAMERAS:
        pts = np.asarray(oracle.polygon(name, *geo, blend), np.int32).reshape(-1, 2)
        mine = oracle.fill_poly(np.zeros((bh, bw), np.uint8), pts) != 0
        im = Image.new("L", (bw, bh), 0)
        ImageDraw.Draw(im).polygon([tuple(int(v) for v in p) for p in pts], fill=255, outline=255)
        pil = np.asarray(im) != 0
        diff = mine != pil
        # interiors agree: a pixel whose 3x3 neighbourhood is uniform in BOTH rasterisations is never in the difference
        def uniform(m):
            p = np.pad(m, 1, mode="edge")
            lo, hi = np.ones(m.shape, bool), np.zeros(m.shape, bool)
            for dy in range(3):
                for dx in range(3):
                    v = p[dy:dy + m.shape[0], dx:dx + m.shape[1]]
                    lo &= v
                    hi |= v
            return lo | ~hi
        assert not np.any(diff & uniform(mine) & uniform(pil)), name
        k = int(np.count_nonzero(diff))
        worst = max(worst, k)
        assert k <= 0.002 * mine.size, f"{name}: {k} boundary pixels differ from PIL"
    print("fillPoly vs PIL.ImageDraw.polygon %s %s: at most %d boundary pixels differ per mask" % (geo, "blend" if blend else "direct", worst))


def test_fisheye_map_inverse_consistency(oracle, repo_rig):
    """cv2.fisheye.initUndistortRectifyMap at surroundBEV.py:99-102: map[i, j] must be the distorted image of undistorted
    pixel (j, i).  The inverse used here (root of the odd polynomial by bracketing) shares nothing with the forward formula."""
    cfg = dict(oracle.DEFAULT_CFG)
    for cam in ("front", "left"):
        K, D, _ = repo_rig.rig[cam]
        D = np.asarray(D, np.float64).ravel()
        fw, fh, ss = cfg["FRAME_WIDTH"], cfg["FRAME_HEIGHT"], cfg["SIZE_SCALE"]
        Kd = oracle.camera_mat_dst(K, fw, fh, cfg["FOCAL_SCALE"], ss)
        m1, m2 = oracle.fisheye_init_undistort_rectify_map(K, D, Kd, (int(fw * ss), int(fh * ss)))
        us, vs = _float_positions(m1, m2)
        rng = np.random.default_rng(11)
        worst = 0.0
        n = 0
        for _ in range(4000):
            i, j = int(rng.integers(0, m1.shape[0])), int(rng.integers(0, m1.shape[1]))
            u, v = us[i, j], vs[i, j]
            if not (0 <= u < fw and 0 <= v < fh):
                continue   # the reference never samples these
            xd, yd = (u - K[0, 2]) / K[0, 0], (v - K[1, 2]) / K[1, 1]
            td = float(np.hypot(xd, yd))
            if td < 1e-9:
                continue
            f = lambda t: t * (1 + D[0] * t ** 2 + D[1] * t ** 4 + D[2] * t ** 6 + D[3] * t ** 8) - td
            theta = optimize.brentq(f, 0.0, np.pi / 2 - 1e-6)
            r = np.tan(theta)
            x, y = xd / td * r, yd / td * r
            jj, ii = Kd[0, 0] * x + Kd[0, 2], Kd[1, 1] * y + Kd[1, 2]
            # the map is quantised to 1/32 pixel in (u, v); pushed back through the inverse that is 1/64 * d(j,i)/d(u,v)
            gain = (1 + r * r) * Kd[0, 0] / K[0, 0]   # radial stretch of the inverse at this radius (upper bound)
            worst = max(worst, np.hypot(jj - j, ii - i) / max(1.0, gain))
            n += 1
        assert n > 1000
        print("fisheye map %s: worst normalised inverse residual %.4f px over %d samples" % (cam, worst, n))
        assert worst < 0.05   # 1/64 pixel of quantisation, normalised by the local stretch


CONDA = "/opt/conda/bin/python3.9"


@pytest.mark.skipif(not os.path.exists(CONDA), reason="no conda python with scikit-image in this image")
def test_bgr2hsv_against_skimage(oracle, tmp_path):
    rng = np.random.default_rng(5)
    bgr = rng.integers(0, 256, (60000, 3), dtype=np.uint8)
    np.save(tmp_path / "bgr.npy", bgr)
    code = ("import numpy as np, json, sys\nfrom skimage.color import rgb2hsv\n"
            "bgr = np.load(sys.argv[1]); hsv = rgb2hsv(bgr[None, :, ::-1].astype(np.float64) / 255.0)[0]\n"
            "np.save(sys.argv[2], hsv)\n")
    r = subprocess.run([CONDA, "-c", code, str(tmp_path / "bgr.npy"), str(tmp_path / "hsv.npy")], capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("skimage not usable: " + r.stderr.strip().splitlines()[-1] if r.stderr.strip() else "skimage not usable")
    ref = np.load(tmp_path / "hsv.npy")
    mine = oracle.bgr2hsv(bgr.reshape(1, -1, 3)).reshape(-1, 3).astype(np.int64)
    v = bgr.max(axis=1).astype(np.int64)
    assert np.array_equal(mine[:, 2], v)
    dS = np.abs(mine[:, 1] - ref[:, 1] * 255.0)
    assert dS.max() <= 1.0 + 1e-9
    chroma = ref[:, 1] > 0
    dH = np.abs(mine[:, 0] - ref[:, 0] * 180.0)
    dH = np.minimum(dH, 180.0 - dH)[chroma]
    assert dH.max() <= 1.0 + 1e-9
    print("BGR2HSV vs skimage: max |dH| %.3f (of 180), max |dS| %.3f (of 255), V exact, %d colours" % (dH.max(), dS.max(), len(bgr)))


@pytest.mark.skipif(not os.path.exists(CONDA), reason="no conda python with scikit-image in this image")
def test_hsv2bgr_against_skimage(oracle, tmp_path):
    """cv2.cvtColor(HSV2BGR) on 8U (surroundBEV.py:76): the float path H * 2 deg, S / 255, V / 255 -> BGR * 255 rounded."""
    rng = np.random.default_rng(6)
    hsv = np.stack([rng.integers(0, 180, 60000), rng.integers(0, 256, 60000), rng.integers(0, 256, 60000)], axis=1).astype(np.uint8)
    np.save(tmp_path / "hsv.npy", hsv)
    code = ("import numpy as np, sys\nfrom skimage.color import hsv2rgb\n"
            "hsv = np.load(sys.argv[1]).astype(np.float64)\n"
            "rgb = hsv2rgb(np.stack([hsv[:, 0] / 180.0, hsv[:, 1] / 255.0, hsv[:, 2] / 255.0], axis=1)[None])[0]\n"
            "np.save(sys.argv[2], rgb)\n")
    r = subprocess.run([CONDA, "-c", code, str(tmp_path / "hsv.npy"), str(tmp_path / "rgb.npy")], capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("skimage not usable")
    ref = np.load(tmp_path / "rgb.npy")[:, ::-1] * 255.0          # -> BGR, 0..255 float
    mine = oracle.hsv2bgr(hsv.reshape(1, -1, 3)).reshape(-1, 3).astype(np.float64)
    d = np.abs(mine - ref)
    print("HSV2BGR vs skimage: max |diff| %.3f LSB, mean %.3f over %d colours" % (d.max(), d.mean(), len(hsv)))
    assert d.max() <= 0.5 + 1e-3          # the float result rounded to the nearest integer


@pytest.mark.skipif(not os.path.exists(CONDA), reason="no conda python with scikit-image in this image")
@pytest.mark.parametrize("f", [0.5, 2.0])
def test_resize_linear_against_skimage(oracle, repo_rig, tmp_path, f):
    """cv2.resize INTER_LINEAR (ScaleImage.__call__, extrinsicCalib.py:125): half-pixel centres, edge replication, 11-bit fixed-point
    weights -- against skimage.transform.resize(order=1) in float64.  Only for factors where the output size is exactly f x the input:
    cv2 samples with scale 1 / fx, skimage with input size / output size, and the two drift apart when f * size is rounded (x0.8 on 512
    columns: up to half a pixel) -- those factors are held against scipy at cv2's own sample positions below."""
    img = repo_rig.image("front")[200:520, 300:812]
    got = oracle.resize_linear(img, f, f).astype(np.float64)
    np.save(tmp_path / "img.npy", img)
    code = ("import numpy as np, sys\nfrom skimage.transform import resize\n"
            "img = np.load(sys.argv[1]).astype(np.float64)\n"
            "out = resize(img, (int(sys.argv[3]), int(sys.argv[4])), order=1, mode='edge', anti_aliasing=False, preserve_range=True)\n"
            "np.save(sys.argv[2], out)\n")
    r = subprocess.run([CONDA, "-c", code, str(tmp_path / "img.npy"), str(tmp_path / "out.npy"), str(got.shape[0]), str(got.shape[1])],
                       capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("skimage not usable")
    ref = np.load(tmp_path / "out.npy")
    d = np.abs(got - ref)
    print("resize x%.1f vs skimage order=1: max |diff| %.2f LSB, mean %.3f" % (f, d.max(), d.mean()))
    assert d.mean() < 0.3 and d.max() <= 1.0   # rounding (0.25 on average) + the 11-bit weight quantisation


@pytest.mark.parametrize("f", [0.37, 0.8, 1.6, 3.3])
def test_resize_linear_against_scipy_at_cv2_positions(oracle, repo_rig, f):
    """the same interpolation at the positions cv2 documents for fx-driven resizes -- source x = (dst x + 0.5) / fx - 0.5, clamped to the
    image (edge replication) -- through scipy's order-1 interpolator in float64"""
    img = repo_rig.image("front")[200:520, 300:812]
    got = oracle.resize_linear(img, f, f).astype(np.float64)
    dh, dw = got.shape[:2]
    ys = np.clip((np.arange(dh) + 0.5) / f - 0.5, 0, img.shape[0] - 1)
    xs = np.clip((np.arange(dw) + 0.5) / f - 0.5, 0, img.shape[1] - 1)
    yy, xx = np.meshgrid(ys, xs, indexing="ij")
    ref = np.stack([ndimage.map_coordinates(img[..., c].astype(np.float64), [yy, xx], order=1, mode="nearest") for c in range(3)], axis=-1)
    d = np.abs(got - ref)
    print("resize x%.2f vs scipy at cv2's positions: max |diff| %.2f LSB, mean %.3f" % (f, d.max(), d.mean()))
    assert d.mean() < 0.3 and d.max() <= 1.0
