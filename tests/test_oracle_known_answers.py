"""Known-answer and cross-statement tests of the CPU oracle.  No GPU.

The oracle is PARITY UNPINNED against real OpenCV (no cv2 in this image, the reference has no tests); these pin it
against hand-derivable answers and against an independent NumPy statement of the same arithmetic.
"""
import numpy as np
import pytest

from oracle import np_twin as T


def ident_maps(w, h, code=0):
    xs, ys = np.meshgrid(np.arange(w), np.arange(h))
    return np.stack([xs, ys], -1).astype(np.int16), np.full((h, w), code, np.uint16)


def test_remap_identity_is_copy(oracle):
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    m1, m2 = ident_maps(53, 37)
    assert np.array_equal(oracle.remap(img, m1, m2), img)


def test_remap_half_pixel_checker(oracle):
    img = np.array([[0, 255], [255, 0]], np.uint8)
    m1 = np.zeros((1, 1, 2), np.int16)
    m2 = np.full((1, 1), 16 * 32 + 16, np.uint16)
    assert oracle.remap(img, m1, m2)[0, 0] == (255 * 256 * 2 + 512) >> 10 == 128


def test_remap_constant_and_border(oracle):
    img = np.full((8, 8, 3), 200, np.uint8)
    m1 = np.array([[[3, 3], [7, 3], [-1, -1], [8, 2], [-2, 0]]], np.int16)
    m2 = np.full((1, 5), 8 * 32 + 8, np.uint16)  # fx = fy = 8/32
    out = oracle.remap(img, m1, m2)
    assert (out[0, 0] == 200).all()
    # right edge: taps x+1 outside -> only the two left taps contribute: weights (24*24 + 24*8) = 768 of 1024
    assert (out[0, 1] == (200 * 768 + 512) >> 10).all()
    # top-left corner from (-1,-1): only the bottom-right tap (8*8 = 64 of 1024)
    assert (out[0, 2] == (200 * 64 + 512) >> 10).all()
    assert (out[0, 3] == 0).all() and (out[0, 4] == 0).all()  # whole 2x2 outside


def test_remap_extra_map2_bits_are_masked(oracle):
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, (16, 16, 3), dtype=np.uint8)
    m1, m2 = ident_maps(16, 16, code=5 * 32 + 7)
    assert np.array_equal(oracle.remap(img, m1, m2), oracle.remap(img, m1, (m2 | 0xFC00).astype(np.uint16)))


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_remap_u8_matches_numpy_twin(oracle, seed):
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 256, (40, 61, 3), dtype=np.uint8)
    m1 = np.stack([rng.integers(-3, 64, (50, 70)), rng.integers(-3, 43, (50, 70))], -1).astype(np.int16)
    m2 = rng.integers(0, 1024, (50, 70)).astype(np.uint16)
    assert np.array_equal(oracle.remap(img, m1, m2), T.remap_u8(img, m1, m2))


@pytest.mark.parametrize("dtype,cn", [(np.int16, 2), (np.uint16, 1)])
def test_remap_f32_matches_numpy_twin(oracle, dtype, cn):
    rng = np.random.default_rng(5)
    lo, hi = (-300, 3000) if dtype == np.int16 else (0, 1024)
    shape = (33, 47, cn) if cn > 1 else (33, 47)
    src = rng.integers(lo, hi, shape).astype(dtype)
    m1 = np.stack([rng.integers(-2, 49, (29, 31)), rng.integers(-2, 35, (29, 31))], -1).astype(np.int16)
    m2 = rng.integers(0, 1024, (29, 31)).astype(np.uint16)
    assert np.array_equal(oracle.remap(src, m1, m2), T.remap_f32(src, m1, m2))


def test_invert_and_perspective_identity_and_shift(oracle):
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (40, 90, 3), dtype=np.uint8)
    assert np.array_equal(oracle.warp_perspective(img, np.eye(3), (90, 40)), img)
    Hs = np.array([[1, 0, 5.0], [0, 1, 3.0], [0, 0, 1]])
    out = oracle.warp_perspective(img, Hs, (90, 40))
    assert np.array_equal(out[3:, 5:], img[:-3, :-5])
    assert (out[:3] == 0).all() and (out[:, :5] == 0).all()


def test_perspective_coords_match_twin_on_repo_homographies(oracle, repo_rig):
    for name, (_, _, H) in repo_rig.rig.items():
        Minv = oracle.invert3x3(H)
        assert np.array_equal(Minv, T.invert3x3(H)), name
        xy, a = oracle.perspective_coords(Minv, (1000, 1000))
        xy2, a2 = T.perspective_coords(Minv, (1000, 1000))
        assert np.array_equal(xy, xy2) and np.array_equal(a, a2), name


def test_fisheye_map_centre_and_twin(oracle, repo_rig):
    K, D, _ = repo_rig.rig["front"]
    # D = 0, K' = K with integer principal point: centre pixel maps to itself, code 0
    K0 = np.array([[350.0, 0, 64.0], [0, 352.0, 48.0], [0, 0, 1]])
    m1, m2 = oracle.fisheye_init_undistort_rectify_map(K0, np.zeros(4), K0, (128, 96))
    assert tuple(m1[48, 64]) == (64, 48) and m2[48, 64] == 0
    # u = fx * atan(r)/r * x + cx along the centre row
    j = 100
    x = (j - 64.0) / 350.0
    u = 350.0 * np.arctan(abs(x)) / abs(x) * x + 64.0
    iu = int(np.rint(u * 32))
    assert m1[48, j, 0] == iu >> 5 and (m2[48, j] & 31) == (iu & 31)
    # the real front camera, reduced size, against the NumPy statement
    Kd = oracle.camera_mat_dst(K, 320, 256, 1.0, 2.0)
    a = oracle.fisheye_init_undistort_rectify_map(K, D, Kd, (640, 512))
    b = T.fisheye_map(K, D, Kd, (640, 512))
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_hsv_primaries_and_grey(oracle):
    px = np.array([[[0, 0, 255], [0, 255, 0], [255, 0, 0], [77, 77, 77], [0, 0, 0], [255, 255, 255]]], np.uint8)
    hsv = oracle.bgr2hsv(px)
    assert hsv[0].tolist() == [[0, 255, 255], [60, 255, 255], [120, 255, 255], [0, 0, 77], [0, 0, 0], [0, 0, 255]]
    assert np.array_equal(oracle.hsv2bgr(hsv), px)


def test_hsv_matches_twin_exhaustive_slice(oracle):
    rng = np.random.default_rng(7)
    px = rng.integers(0, 256, (512, 512, 3), dtype=np.uint8)
    hsv = oracle.bgr2hsv(px)
    assert np.array_equal(hsv, T.bgr2hsv(px))
    assert np.array_equal(oracle.hsv2bgr(hsv), T.hsv2bgr(hsv))
    assert hsv[..., 0].max() < 180


def test_hsv_roundtrip_close_to_colorsys():
    import colorsys
    from oracle import oracle as O

    rng = np.random.default_rng(8)
    px = rng.integers(0, 256, (1, 200, 3), dtype=np.uint8)
    back = O.hsv2bgr(O.bgr2hsv(px))
    assert np.abs(back.astype(int) - px.astype(int)).max() <= 6  # lossy by a few LSB, never wild
    for (b, g, r), (h, s, v) in zip(px[0].tolist(), O.bgr2hsv(px)[0].tolist()):
        hh, ss, vv = colorsys.rgb_to_hsv(r / 255, g / 255, b / 255)
        assert v == max(b, g, r) and abs(s - ss * 255) <= 1.0
        if ss > 0.1:
            assert min(abs(h - hh * 180), 180 - abs(h - hh * 180)) <= 1.5


def test_luminance_shift_and_saturation(oracle):
    a = np.full((4, 4, 3), 250, np.uint8)
    b = np.full((4, 4, 3), 10, np.uint8)
    out = oracle.luminance_balance([a, a, b, b])  # mean V = 130: shifts -120, -120, +120, +120
    assert (out[0] == 130).all() and (out[2] == 130).all()
    hi = np.full((2, 2, 3), 255, np.uint8)
    lo = np.zeros((2, 2, 3), np.uint8)
    out = oracle.luminance_balance([hi, lo, lo, lo])  # deltas -191, +64 (63.75 rounds to 64)
    assert (out[0] == 64).all() and (out[1] == 64).all()


def test_fill_poly_rectangle_inclusive(oracle):
    m = oracle.fill_poly(np.zeros((20, 30), np.uint8), [[3, 4], [12, 4], [12, 9], [3, 9]])
    ref = np.zeros((20, 30), np.uint8)
    ref[4:10, 3:13] = 255
    assert np.array_equal(m, ref)


def test_fill_poly_clips_offscreen_vertices(oracle):
    m = oracle.fill_poly(np.zeros((10, 10), np.uint8), [[0, 0], [10, 0], [10, 10], [0, 10]])
    assert (m == 255).all()
    m = oracle.fill_poly(np.zeros((10, 10), np.uint8), [[-5, -5], [4, -5], [4, 4], [-5, 4]])
    assert (m[:5, :5] == 255).all() and m.sum() == 25 * 255


def test_fill_poly_triangle_contains_bresenham_boundary_and_interior(oracle):
    from matplotlib.path import Path

    pts = np.array([[2, 1], [47, 9], [20, 38]])
    m = oracle.fill_poly(np.zeros((40, 50), np.uint8), pts)
    ys, xs = np.mgrid[0:40, 0:50]
    inside = Path(pts).contains_points(np.c_[xs.ravel(), ys.ravel()], radius=-1e-9).reshape(40, 50)
    assert (m[inside] == 255).all()  # strict interior is always filled
    grown = Path(pts).contains_points(np.c_[xs.ravel(), ys.ravel()], radius=1.6).reshape(40, 50) | inside
    assert not (m[~grown] != 0).any()  # nothing further than ~1 px outside the outline
    for vx, vy in pts:
        assert m[vy, vx] == 255


def test_direct_masks_share_their_seam_and_skip_the_car(oracle):
    geo = (1000, 1000, 250, 400)
    masks = {n: oracle.direct_mask(n, *geo) for n in ("front", "back", "left", "right")}
    cover = sum((m != 0).astype(int) for m in masks.values())
    assert cover.max() == 2  # seam pixels belong to two cameras (reference behaviour: saturating double add)
    assert (cover[301:699, 376:624] == 0).all()  # car rectangle interior in no mask
    assert masks["front"][0, 0] and masks["left"][0, 0]
    # mirror symmetry of the rig geometry: front/back and left/right flip onto each other except rasterisation ties
    assert abs(int((masks["front"] != 0).sum()) - int((masks["back"] != 0).sum())) < 2000


def test_blend_weights_bounds_and_seam_zero(oracle):
    geo = (200, 200, 50, 80)
    mf = oracle.blend_mask_for("front", *geo)
    ml = oracle.blend_mask_for("left", *geo)
    both = (mf != 0) & (ml != 0)
    s = mf.astype(int) + ml.astype(int)
    assert both.any() and set(np.unique(s[both])) <= {254, 255}
    fl = oracle.seam("FL", *geo)
    assert mf[fl[0][1], fl[0][0]] == 0  # a point on front's own seam gets weight 0 for front
    w = oracle.blend_weight(mf)
    assert w.dtype == np.float32 and w.shape == (200, 200, 3) and w.max() == 1.0


def test_stitch_arithmetic(oracle):
    a = np.array([[[250, 3, 128]]], np.uint8)
    b = np.array([[[10, 4, 128]]], np.uint8)
    assert oracle.add_sat(a, b).tolist() == [[[255, 7, 255]]]
    img = np.zeros((2, 2, 3), np.uint8)
    img[..., 0], img[..., 1], img[..., 2] = 50, 100, 150
    out = oracle.color_balance(img)  # K = 100: gains 2, 1, 2/3
    assert (out[..., 0] == 100).all() and (out[..., 1] == 100).all() and (out[..., 2] == 100).all()
    car = np.arange(12, dtype=np.uint8).reshape(2, 2, 3)
    assert np.array_equal(oracle.padding(car, 5, 4)[1:3, 1:3], car) and oracle.padding(car, 5, 4).sum() == car.sum()


def test_weight_mul_truncates_all_pairs(oracle):
    from oracle.oracle import lib, _p

    # every (mask value, pixel value) pair against numpy evaluating the reference expression itself
    # (img * weight_f32).astype(np.uint8)  (surroundBEV.py:187-188, 279-280)
    mask = np.repeat(np.arange(256, dtype=np.uint8), 256)
    img = np.tile(np.arange(256, dtype=np.uint8), 256)
    w = (mask / 255.0).astype(np.float32)
    out = np.empty(img.size, np.uint8)
    lib().orc_weight_mul(_p(img), _p(w), img.size, _p(out))
    assert np.array_equal(out, (img * w).astype(np.uint8))
    assert out[255 * 256 + 255] == 255 and out[128 * 256 + 255] == 128


def test_pinhole_map_identity_and_radial(oracle):
    """cv2.initUndistortRectifyMap known answers: D = 0 and K' = K is the identity map; a pure k1 term moves a pixel
    radially by x * k1 * r^2 (hand-computed), and the fixed-point split is the same Q5 packing as the fisheye map."""
    K = np.array([[400.0, 0, 320.0], [0, 410.0, 240.0], [0, 0, 1]])
    m1, m2 = oracle.init_undistort_rectify_map(K, np.zeros(5), K, (640, 480))
    xs, ys = np.meshgrid(np.arange(640), np.arange(480))
    # iR = cofactor inverse of K: u = fx * ((j - cx) / fx) + cx may be off by an ulp, far below 1/32 px
    assert np.array_equal(m1[..., 0], xs) and np.array_equal(m1[..., 1], ys) and (m2 == 0).all()
    k1 = 0.1
    m1, m2 = oracle.init_undistort_rectify_map(K, [k1, 0, 0, 0, 0], K, (640, 480))
    j, i = 600, 100
    x, y = (j - 320.0) / 400.0, (i - 240.0) / 410.0
    r2 = x * x + y * y
    u, v = 400.0 * x * (1 + k1 * r2) + 320.0, 410.0 * y * (1 + k1 * r2) + 240.0
    iu, iv = int(np.rint(u * 32)), int(np.rint(v * 32))
    assert (m1[i, j, 0], m1[i, j, 1]) == (iu >> 5, iv >> 5)
    assert m2[i, j] == (iv & 31) * 32 + (iu & 31)
    # tangential terms: p1 shifts x by 2 p1 x y and y by p1 (r2 + 2 y2)
    p1 = 0.01
    m1, m2 = oracle.init_undistort_rectify_map(K, [0, 0, p1, 0, 0], K, (640, 480))
    u = 400.0 * (x + p1 * 2 * x * y) + 320.0
    v = 410.0 * (y + p1 * (r2 + 2 * y * y)) + 240.0
    iu, iv = int(np.rint(u * 32)), int(np.rint(v * 32))
    assert (m1[i, j, 0], m1[i, j, 1]) == (iu >> 5, iv >> 5)


def test_translate_known_answers(oracle):
    """cv2.warpAffine with an integer shift (CenterImage.translate, extrinsicCalib.py:54-59): a copy moved by
    (shift_x, shift_y), zeros where the source tap falls outside."""
    img = np.arange(5 * 7 * 3, dtype=np.uint8).reshape(5, 7, 3)
    assert np.array_equal(oracle.translate(img, 0, 0), img)
    out = oracle.translate(img, 2, 1)
    assert np.array_equal(out[1:, 2:], img[:-1, :-2]) and not out[0].any() and not out[:, :2].any()
    out = oracle.translate(img, -3, -2)
    assert np.array_equal(out[:-2, :-3], img[2:, 3:]) and not out[-2:].any() and not out[:, -3:].any()
    assert not oracle.translate(img, 7, 0).any() and not oracle.translate(img, 0, -5).any()


def test_resize_known_answers(oracle):
    """cv2.resize INTER_LINEAR on 8U (ScaleImage.__call__, extrinsicCalib.py:125): hand-derived cases."""
    rng = np.random.default_rng(11)
    img = rng.integers(0, 256, (12, 18, 3), dtype=np.uint8)
    # scale 1: weights (2048, 0) both ways -> ((2048 * (S*2048 >> 4)) >> 16) = 4 S -> (4 S + 2) >> 2 = S
    assert np.array_equal(oracle.resize_linear(img, 1.0, 1.0), img)
    # scale 1/2: every tap pair has weight 1024 -> (a + b + c + d + 2) >> 2, which is also OpenCV's area-fast shortcut
    half = oracle.resize_linear(img, 0.5, 0.5)
    i32 = img.astype(np.int32)
    want = (i32[0::2, 0::2] + i32[0::2, 1::2] + i32[1::2, 0::2] + i32[1::2, 1::2] + 2) >> 2
    assert half.shape == (6, 9, 3) and np.array_equal(half, want.astype(np.uint8))
    # dsize = cvRound(size * f): round-half-to-even
    assert oracle.resize_linear(np.zeros((5, 7, 3), np.uint8), 0.5, 0.5).shape == (2, 4, 3)   # 2.5 -> 2, 3.5 -> 4
    # 2x upscale of the row [0, 100] (one image row): columns sample at -0.25, 0.25, 0.75, 1.25
    row = np.zeros((1, 2, 3), np.uint8)
    row[0, 1] = 100
    up = oracle.resize_linear(row, 2.0, 2.0)
    # x=0: s<0 -> 0; x=1: f=.25 -> S=100*512=51200, (51200>>4)=3200, rows both clip to row 0 with b=(512,1536):
    #   ((512*3200)>>16) + ((1536*3200)>>16) = 25 + 75 = 100 -> (100+2)>>2 = 25;  x=2: f=.75 -> 75;  x=3: s>=w-1 -> 100
    assert up.shape == (2, 4, 3) and up[0, :, 0].tolist() == [0, 25, 75, 100] and np.array_equal(up[0], up[1])
    # constant images stay constant up to the fixed-point floor of the two vertical terms (never above the value)
    for v in (1, 77, 255):
        c = oracle.resize_linear(np.full((9, 11, 3), v, np.uint8), 1.7, 1.7)
        assert c.max() <= v and c.min() >= v - 1


@pytest.mark.parametrize("f", [0.37, 0.8, 1.3, 2.6])
def test_resize_close_to_float_bilinear(oracle, f):
    """Independent check of the geometry (half-pixel centres, edge replication): float bilinear within 1 LSB."""
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (23, 31, 3), dtype=np.uint8)
    got = oracle.resize_linear(img, f, f).astype(np.float64)
    dh, dw = got.shape[:2]
    ys = np.clip((np.arange(dh) + 0.5) / f - 0.5, 0, img.shape[0] - 1)
    xs = np.clip((np.arange(dw) + 0.5) / f - 0.5, 0, img.shape[1] - 1)
    y0, x0 = np.floor(ys).astype(int), np.floor(xs).astype(int)
    y1, x1 = np.minimum(y0 + 1, img.shape[0] - 1), np.minimum(x0 + 1, img.shape[1] - 1)
    wy, wx = (ys - y0)[:, None, None], (xs - x0)[None, :, None]
    I = img.astype(np.float64)
    want = (I[y0][:, x0] * (1 - wx) + I[y0][:, x1] * wx) * (1 - wy) + (I[y1][:, x0] * (1 - wx) + I[y1][:, x1] * wx) * wy
    assert np.abs(got - want).max() <= 1.0
