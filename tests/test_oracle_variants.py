"""The OpenCV-version-sensitive choices of the oracle are SWITCHES (oracle.set_variant, same keys as bevw_set_compat of
include/bevwarp.h), so that a golden file from a real cv2 (tests/golden/README.md) decides them instead of the author.
These CPU tests pin what each switch changes and how much:

  fillPoly  (surroundBEV.py:159,234)  OpenCV >= 4.5.2 edge collection vs 2.4 .. 4.5.1: a few dozen to a few hundred SEAM
            pixels per mask (of 10^6), every one of them on the polygon boundary -- one wrong seam pixel is >> 1 LSB in the
            stitched image, which is why SURVEY.md A.5 calls it the highest-risk item;
  addWeighted (surroundBEV.py:52-54)  CV_64F vs CV_32F evaluation of sat_u8(cvRound(ch * k)): identical on the reference's
            sample frame, at most 1 LSB on < 1e-4 of all (value, gain) pairs.
"""
import numpy as np
import pytest

GEOMETRIES = [(1000, 1000, 250, 400), (1000, 1000, 200, 350), (1080, 1080, 270, 432), (999, 801, 251, 333), (640, 480, 100, 160)]


@pytest.fixture(autouse=True)
def restore_variants(oracle):
    yield
    oracle.set_variant(oracle.VARIANT_FILLPOLY, 1)
    oracle.set_variant(oracle.VARIANT_ADDWEIGHTED, 1)
    oracle.set_variant(oracle.VARIANT_WARP, 0)
    oracle.set_variant(oracle.VARIANT_REMAP, 0)


def test_defaults_and_round_trip(oracle):
    assert oracle.get_variant(oracle.VARIANT_FILLPOLY) == 1 and oracle.get_variant(oracle.VARIANT_ADDWEIGHTED) == 1
    for key in (oracle.VARIANT_FILLPOLY, oracle.VARIANT_ADDWEIGHTED):
        oracle.set_variant(key, 0)
        assert oracle.get_variant(key) == 0
        oracle.set_variant(key, 1)
        assert oracle.get_variant(key) == 1
    assert oracle.get_variant(7) == -1


def _masks(oracle, geo, blend):
    return [(oracle.blend_mask_for(n, *geo) if blend else oracle.direct_mask(n, *geo)) for n in oracle.CAMERAS]


def _boundary(mask):
    """pixels whose 3x3 neighbourhood holds both a zero and a non-zero value"""
    m = np.pad(mask != 0, 1, mode="edge")
    lo = np.ones(mask.shape, bool)
    hi = np.zeros(mask.shape, bool)
    for dy in range(3):
        for dx in range(3):
            v = m[dy:dy + mask.shape[0], dx:dx + mask.shape[1]]
            lo &= v
            hi |= v
    return hi & ~lo


@pytest.mark.parametrize("geo", GEOMETRIES)
@pytest.mark.parametrize("blend", [False, True])
def test_fillpoly_variants_differ_only_on_the_boundary(oracle, geo, blend):
    oracle.set_variant(oracle.VARIANT_FILLPOLY, 1)
    new = _masks(oracle, geo, blend)
    oracle.set_variant(oracle.VARIANT_FILLPOLY, 0)
    old = _masks(oracle, geo, blend)
    total = 0
    for n, a, b in zip(oracle.CAMERAS, new, old):
        diff = (a != 0) != (b != 0)   # blend masks carry weights: compare coverage
        k = int(np.count_nonzero(diff))
        total += k
        assert k <= 0.001 * a.size, f"{n}: {k} pixels differ"
        # every differing pixel sits on the polygon boundary of BOTH variants (interior and exterior agree)
        assert not np.any(diff & ~(_boundary(a) | _boundary(b))), n
    # the switch is not a no-op: some seam pixel of the rig changes (this is what a cv2 golden decides)
    assert total > 0


def test_fillpoly_axis_aligned_rectangle_is_variant_free(oracle):
    """a rectangle fills its inclusive bounds -- in both variants."""
    for v in (0, 1):
        oracle.set_variant(oracle.VARIANT_FILLPOLY, v)
        m = oracle.fill_poly(np.zeros((40, 50), np.uint8), np.array([[5, 7], [30, 7], [30, 22], [5, 22]], np.int32))
        want = np.zeros((40, 50), np.uint8)
        want[7:23, 5:31] = 255
        assert np.array_equal(m, want), v


def test_addweighted_variants_on_the_sample_frame(oracle, repo_rig):
    img = repo_rig.image("back")
    oracle.set_variant(oracle.VARIANT_ADDWEIGHTED, 1)
    a = oracle.color_balance(img.copy())
    oracle.set_variant(oracle.VARIANT_ADDWEIGHTED, 0)
    b = oracle.color_balance(img.copy())
    assert np.array_equal(a, b)   # the reference's own data does not tell the two work types apart


def test_addweighted_variants_exhaustive_values(oracle):
    """all 256 byte values x 3000 gains around 1: the two work types differ by at most 1 LSB, on < 1e-4 of the pairs."""
    rng = np.random.default_rng(7)
    vals = np.repeat(np.arange(256, dtype=np.uint8), 3).reshape(256, 3)
    ndiff = ntot = 0
    worst = 0
    L = oracle.lib()
    for _ in range(1000):
        gains = np.ascontiguousarray(rng.uniform(0.6, 1.6, 3))
        out = []
        for v in (1, 0):
            oracle.set_variant(oracle.VARIANT_ADDWEIGHTED, v)
            img = vals.copy()
            L.orc_gain(img.ctypes.data, 256, gains.ctypes.data)
            out.append(img.astype(np.int32))
        d = np.abs(out[0] - out[1])
        worst = max(worst, int(d.max()))
        ndiff += int(np.count_nonzero(d))
        ntot += d.size
    assert worst <= 1
    assert ndiff <= 1e-4 * ntot, (ndiff, ntot)


# ---- VARIANT_WARP / VARIANT_REMAP: candidates for OpenCV >= 4.11's float32 linear kernels (bevoracle.c A.4b) -----------------------
def test_warp_family_known_answers(oracle):
    """Every member: the identity homography copies, an integer translation shifts with a zero border -- positions that
    are exact in float32 leave no room for the members to differ -- for 8UC3, 8UC1 and 16UC1 images; the two-channel 16S map (not a type
    the float kernels take) keeps the classic path whatever the switch says."""
    rng = np.random.default_rng(3)
    u8 = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    u16 = rng.integers(0, 65536, (37, 53), dtype=np.uint16)
    s16 = rng.integers(-3000, 3000, (37, 53, 2), dtype=np.int16)
    shift = np.array([[1, 0, 5], [0, 1, -3], [0, 0, 1.0]])
    classic_s16 = oracle.warp_perspective(s16, shift @ np.diag([1.01, 0.99, 1.0]), (60, 40))
    assert oracle.get_variant(oracle.VARIANT_WARP) == 0 and oracle.get_variant(oracle.VARIANT_REMAP) == 0
    for m in oracle.WARP_FAMILY:
        oracle.set_variant(oracle.VARIANT_WARP, m)
        for img in (u8, u8[:, :, 0], u16):
            assert np.array_equal(oracle.warp_perspective(img, np.eye(3), (53, 37)), img), m
            got = oracle.warp_perspective(img, shift, (53, 37))
            want = np.zeros_like(img)
            want[:-3, 5:] = img[3:, :-5]
            assert np.array_equal(got, want), m
        assert np.array_equal(oracle.warp_perspective(s16, shift @ np.diag([1.01, 0.99, 1.0]), (60, 40)), classic_s16), m


def test_warp_family_members_stay_within_one_lsb_of_each_other_and_of_the_classic_path_on_smooth_images(oracle):
    """On a smooth image the members differ from each other by rounding (<= 1 LSB) and from the classic kernels by the 1/32-pixel
    position quantisation times the gradient (<= 1 LSB at a gradient of <= 16 / pixel): what BASELINE's +-1 LSB needs from a user on
    OpenCV >= 4.11, whichever member it is."""
    y, x = np.mgrid[0:200, 0:260].astype(np.float64)
    smooth = np.stack([(x * 0.9 + y * 0.2) % 256, 128 + 100 * np.sin(x / 40) * np.cos(y / 35), (x + y) / 2], -1).clip(0, 255).astype(np.uint8)
    smooth[:, :, 0] = (x * 0.9 + y * 0.2).clip(0, 255).astype(np.uint8)   # (no wrap: a sawtooth edge is not smooth)
    H = np.linalg.inv(np.array([[0.75, 0.05, 20.0], [-0.04, 0.8, 30.0], [1.0e-4, -0.5e-4, 1.0]]))   # every destination pixel samples the interior
    classic = oracle.warp_perspective(smooth, H, (240, 180)).astype(np.int32)
    outs = []
    for m in oracle.WARP_FAMILY:
        oracle.set_variant(oracle.VARIANT_WARP, m)
        outs.append(oracle.warp_perspective(smooth, H, (240, 180)).astype(np.int32))
    inner = (slice(None), slice(None))
    for o in outs:
        assert np.abs(o[inner] - outs[0][inner]).max() <= 1
        assert np.abs(o[inner] - classic[inner]).max() <= 1
    assert any(not np.array_equal(o, outs[0]) for o in outs)   # the members ARE different kernels


def test_remap_tie_rule(oracle):
    """(S + 512) >> 10 against round-half-even of S / 1024: identical unless S = 512 (mod 1024) with an odd rounded-up result."""
    m1 = np.zeros((3, 3, 2), np.int16)
    m2 = np.full((3, 3), 16 * 32 + 16, np.uint16)            # fx = fy = 1/2: weights 256 each
    for taps, up, even in [((1, 1, 0, 0), 1, 0), ((3, 3, 0, 0), 2, 2), ((1, 0, 0, 0), 0, 0), ((2, 2, 1, 1), 2, 2), ((255, 255, 255, 254), 255, 255),
                           ((5, 0, 0, 0), 1, 1), ((2, 0, 0, 0), 1, 0), ((6, 0, 0, 0), 2, 2)]:
        src = np.zeros((5, 5), np.uint8)
        src[0, 0], src[0, 1], src[1, 0], src[1, 1] = taps
        oracle.set_variant(oracle.VARIANT_REMAP, 0)
        assert oracle.remap(src, m1, m2)[0, 0] == up, taps
        oracle.set_variant(oracle.VARIANT_REMAP, 1)
        assert oracle.remap(src, m1, m2)[0, 0] == even, taps
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (64, 64, 3), dtype=np.uint8)
    mm1 = rng.integers(0, 62, (50, 50, 2)).astype(np.int16)
    mm2 = rng.integers(0, 1024, (50, 50)).astype(np.uint16)
    oracle.set_variant(oracle.VARIANT_REMAP, 0)
    a = oracle.remap(img, mm1, mm2).astype(np.int32)
    oracle.set_variant(oracle.VARIANT_REMAP, 1)
    b = oracle.remap(img, mm1, mm2).astype(np.int32)
    assert (a - b).min() >= 0 and (a - b).max() <= 1 and np.count_nonzero(a - b) < 0.01 * a.size
