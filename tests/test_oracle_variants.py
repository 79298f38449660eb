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
