"""Oracle vs a REAL OpenCV, through tests/golden/cv2_goldens.npz (written by tests/golden/make_goldens_with_cv2.py on a
machine that has cv2 and the reference).  The file does not exist yet -- no cv2 in this image -- so these tests skip and
DESIGN.md keeps saying "parity unpinned"; the day the file is committed they become the pin."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import _golden_cases as GC  # noqa: E402

PATH = os.environ.get("BEVW_GOLDENS_FILE", os.path.join(ROOT, "tests", "golden", "cv2_goldens.npz"))
# BEVW_REQUIRE_GOLDENS=1: whoever has just run make_goldens_with_cv2.py wants a red or a green answer per case, never a skip -- a missing
# file and a case the file does not hold both FAIL under the flag
REQUIRE = os.environ.get("BEVW_REQUIRE_GOLDENS", "0") not in ("", "0")
pytestmark = pytest.mark.skipif(not os.path.exists(PATH) and not REQUIRE, reason="no cv2 goldens (tests/golden/make_goldens_with_cv2.py)")


@pytest.fixture(scope="module")
def goldens():
    if not os.path.exists(PATH):
        pytest.fail("BEVW_REQUIRE_GOLDENS is set and %s does not exist (tests/golden/README.md)" % PATH)
    return np.load(PATH)


def check(goldens, name, arr):
    if name + "__sha" not in goldens.files:
        if REQUIRE:
            pytest.fail("case %s is not in the golden file (regenerate it with the current make_goldens_with_cv2.py)" % name)
        pytest.skip("case %s not in the golden file" % name)
    assert tuple(goldens[name + "__shape"]) == arr.shape, name
    if str(goldens[name + "__sha"]) != GC.digest(arr):
        d = np.abs(GC.probe(arr).astype(np.int64) - goldens[name + "__probe"].astype(np.int64))
        pytest.fail("%s differs from OpenCV %s: probe max |diff| %d, %d of %d probe entries differ" %
                    (name, goldens["cv2_version"], int(d.max()), int(np.count_nonzero(d)), d.size))


@pytest.fixture(scope="module", autouse=True)
def variants_decided_by_the_goldens(goldens, oracle, repo_rig):
    """The golden file DECIDES the OpenCV-version-sensitive switches (oracle.set_variant / bevw_set_compat): the variant whose
    direct masks / colour balance reproduce the cv2 outputs is selected for the rest of this module and printed, so that the
    defaults in oracle/bevoracle.c and csrc/bevwarp.hip can be flipped to match the OpenCV that produced the file."""
    chosen = {}
    geo = (1000, 1000, 250, 400)
    for v in (1, 0):
        oracle.set_variant(oracle.VARIANT_FILLPOLY, v)
        ok = all(str(goldens["mask_direct_%s__sha" % n]) == GC.digest(oracle.direct_mask(n, *geo)) for n in GC.CAMS
                 if "mask_direct_%s__sha" % n in goldens.files)
        if ok:
            chosen["fillPoly"] = v
            break
    for v in (1, 0):
        oracle.set_variant(oracle.VARIANT_ADDWEIGHTED, v)
        if "color_balance_back__sha" in goldens.files and \
                str(goldens["color_balance_back__sha"]) == GC.digest(oracle.color_balance(repo_rig.image("back"))):
            chosen["addWeighted"] = v
            break
    # warpPerspective / remap IMPLEMENTATION (the probes stored whole): the classic kernels first, then every member of the float32 family
    # (oracle.WARP_FAMILY, bevoracle.c A.4b: candidates for OpenCV >= 4.11); a member is selected only if it reproduces BOTH the 8UC3 and
    # the 16UC1 probe element for element.  The remap tie rule likewise on probe_remap_8uc3.
    if "probe_warp_8uc3__full" in goldens.files and "probe_warp_16uc1__full" in goldens.files:
        u16, _s16, u8 = GC.probe_images()
        Hp = np.array(GC.PROBE_H)
        for m in [0] + list(oracle.WARP_FAMILY):
            oracle.set_variant(oracle.VARIANT_WARP, m)
            if np.array_equal(goldens["probe_warp_8uc3__full"], oracle.warp_perspective(u8, Hp, GC.PROBE_DSIZE)) and \
                    np.array_equal(goldens["probe_warp_16uc1__full"], oracle.warp_perspective(u16, Hp, GC.PROBE_DSIZE)):
                chosen["warpPerspective"] = m
                break
    if "probe_remap_8uc3__full" in goldens.files:
        _u16, _s16, u8 = GC.probe_images()
        pm1, pm2 = GC.probe_maps()
        for v in (0, 1):
            oracle.set_variant(oracle.VARIANT_REMAP, v)
            if np.array_equal(goldens["probe_remap_8uc3__full"], oracle.remap(u8, pm1, pm2)):
                chosen["remap"] = v
                break
    oracle.set_variant(oracle.VARIANT_FILLPOLY, chosen.get("fillPoly", 1))
    oracle.set_variant(oracle.VARIANT_ADDWEIGHTED, chosen.get("addWeighted", 1))
    oracle.set_variant(oracle.VARIANT_WARP, chosen.get("warpPerspective", 0))
    oracle.set_variant(oracle.VARIANT_REMAP, chosen.get("remap", 0))
    print("cv2 %s selects variants %s (fillPoly / addWeighted: 1 = the shipped default; warpPerspective: %s; remap: 0 = half up, the shipped "
          "default; a key that is missing matched NO candidate)" % (goldens["cv2_version"], chosen, oracle.warp_mode_name(chosen.get("warpPerspective", 0))))
    yield chosen
    oracle.set_variant(oracle.VARIANT_FILLPOLY, 1)
    oracle.set_variant(oracle.VARIANT_ADDWEIGHTED, 1)
    oracle.set_variant(oracle.VARIANT_WARP, 0)
    oracle.set_variant(oracle.VARIANT_REMAP, 0)


@pytest.fixture(scope="module")
def ref_default(oracle, repo_rig):
    return {b: oracle.RefBevGenerator(repo_rig.rig, dict(oracle.DEFAULT_CFG), blend=b, balance=False) for b in (False, True)}


@pytest.mark.parametrize("cam", range(4))
def test_tables_masks_and_single_camera_remaps(goldens, ref_default, repo_rig, cam):
    n = GC.CAMS[cam]
    r = ref_default[False]
    check(goldens, "und_map1_" + n, r.cameras[cam].undistort_maps[0])
    check(goldens, "und_map2_" + n, r.cameras[cam].undistort_maps[1])
    check(goldens, "bev_map1_" + n, r.cameras[cam].bev_maps[0])
    check(goldens, "bev_map2_" + n, r.cameras[cam].bev_maps[1])
    check(goldens, "mask_direct_" + n, r.masks[cam])
    check(goldens, "mask_blend_" + n, ref_default[True].masks[cam])
    check(goldens, "raw2bev_" + n, r.cameras[cam].raw2bev(repo_rig.frames()[cam]))
    if cam == 0:
        check(goldens, "undistort_front", r.cameras[0].undistort(repo_rig.frames()[0]))


@pytest.mark.parametrize("blend,balance", GC.MODES)
def test_bev_generator_modes(goldens, oracle, repo_rig, blend, balance):
    cfg = dict(oracle.DEFAULT_CFG, CAR_WIDTH=GC.MAIN_CAR[0], CAR_HEIGHT=GC.MAIN_CAR[1])
    ref = oracle.RefBevGenerator(repo_rig.rig, cfg, blend=blend, balance=balance)
    car = oracle.padding(repo_rig.image("car"), cfg["BEV_WIDTH"], cfg["BEV_HEIGHT"])
    check(goldens, "car_padded", car)
    tag = "bev_%d%d" % (blend, balance)
    check(goldens, tag, ref(*repo_rig.frames()))
    check(goldens, tag + "_car", ref(*repo_rig.frames(), car=car))


def test_balance_helpers(goldens, oracle, repo_rig):
    for n, out in zip(GC.CAMS, oracle.luminance_balance(repo_rig.frames())):
        check(goldens, "lum_" + n, out)
    check(goldens, "color_balance_back", oracle.color_balance(repo_rig.image("back")))


def test_addweighted_scalar_work_type(goldens, oracle, variants_decided_by_the_goldens):
    """cv2.addWeighted(ramp, gain, 0, 0, 0) with the scalar second source of surroundBEV.py:52-54 on three gains where a CV_32F and a
    CV_64F evaluation differ: decides the `addWeighted` switch on its own, whatever the image-based case selected."""
    ramp = np.repeat(np.arange(256, dtype=np.uint8)[:, None, None], 3, axis=2).copy()    # [256, 1, 3]: one gain per channel
    hits = {}
    for v in (1, 0):
        oracle.set_variant(oracle.VARIANT_ADDWEIGHTED, v)
        img = ramp.copy()
        oracle.lib().orc_gain(img.ctypes.data, 256, np.array(GC.ADDWEIGHTED_GAINS, np.float64).ctypes.data)
        hits[v] = [("addweighted_scalar_%d__sha" % k) in goldens.files and
                   str(goldens["addweighted_scalar_%d__sha" % k]) == GC.digest(np.ascontiguousarray(img[:, :, k])) for k in range(3)]
    oracle.set_variant(oracle.VARIANT_ADDWEIGHTED, variants_decided_by_the_goldens.get("addWeighted", 1))
    if not any(("addweighted_scalar_%d__sha" % k) in goldens.files for k in range(3)):
        if REQUIRE:
            pytest.fail("the golden file has no addweighted_scalar_* cases (regenerate it)")
        pytest.skip("no addweighted_scalar_* cases in the golden file")
    assert all(hits[1]) or all(hits[0]), "neither work type reproduces cv2.addWeighted with a scalar src2: %s" % hits
    want = 1 if all(hits[1]) else 0
    assert variants_decided_by_the_goldens.get("addWeighted", want) == want, "image-based and ramp-based cases disagree on the work type"


def test_calibrator_paths(goldens, oracle, repo_rig):
    K, D = repo_rig.rig["front"][0], repo_rig.rig["front"][1]
    src = repo_rig.image("incalib")
    h, w = src.shape[:2]
    Kd = oracle.camera_mat_dst(K, w, h, 0.5, 1)
    m1, m2 = oracle.fisheye_init_undistort_rectify_map(K, D, Kd, (w, h))
    check(goldens, "incalib_undistort", oracle.remap(src, m1, m2))
    p1, p2 = oracle.init_undistort_rectify_map(K, list(GC.PINHOLE_D), Kd, (w, h))
    check(goldens, "pinhole_map1", p1)
    check(goldens, "pinhole_map2", p2)
    check(goldens, "excalib_warp", oracle.warp_perspective(repo_rig.image("excalib_src"), repo_rig.rig["back"][2], (1000, 1000)))
    small = np.ascontiguousarray(repo_rig.image("back")[:301, :403])
    check(goldens, "translate", oracle.translate(small, 403 // 2 - 100, 301 // 2 - 250))
    for f in GC.RESIZE_FACTORS:
        check(goldens, "resize_%g" % f, oracle.resize_linear(small, f, f))


def test_implementation_probes(goldens, oracle):
    """Which warpPerspective / remap IMPLEMENTATION the golden file's OpenCV runs.  The oracle restates the classic kernels (fixed-point
    5-bit x 5-bit weights for 8U sources, float weights for 16-bit sources: every OpenCV from 2.4 to 4.10); OpenCV >= 4.11 ships new
    warpPerspective kernels (8U / 16U / 32F) and a reworked remap, for which the oracle and the engine carry a family of CANDIDATES
    (VARIANT_WARP / BEVW_COMPAT_WARP, VARIANT_REMAP / BEVW_COMPAT_REMAP) -- the module fixture has selected the member that reproduces
    the probes, if one does, before this test runs.  The probes are stored whole, so a mismatch is reported with its
    size: how many elements differ and by how much -- 'classic' (0 differences) or 'another implementation' (sub-LSB rounding
    differences on a large share of the pixels), not just 'differs'."""
    names = [n for n in ("probe_warp_16uc1", "probe_warp_16sc2", "probe_warp_8uc3", "probe_remap_8uc3") if n + "__full" in goldens]
    if not names:
        if os.environ.get("BEVW_REQUIRE_GOLDENS"):
            pytest.fail("the golden file has no probe_* cases (regenerate it with the current make_goldens_with_cv2.py)")
        pytest.skip("no probe_* cases in the golden file")
    u16, s16, u8 = GC.probe_images()
    Hp = np.array(GC.PROBE_H)
    pm1, pm2 = GC.probe_maps()
    mine = {"probe_warp_16uc1": oracle.warp_perspective(u16, Hp, GC.PROBE_DSIZE), "probe_warp_16sc2": oracle.warp_perspective(s16, Hp, GC.PROBE_DSIZE),
            "probe_warp_8uc3": oracle.warp_perspective(u8, Hp, GC.PROBE_DSIZE), "probe_remap_8uc3": oracle.remap(u8, pm1, pm2)}
    report = []
    for n in names:
        want = goldens[n + "__full"]
        d = np.abs(want.astype(np.int64) - mine[n].astype(np.int64))
        report.append("%s: %d of %d elements differ, max |diff| %d" % (n, int(np.count_nonzero(d)), d.size, int(d.max())))
    print("OpenCV %s implementation probes: %s" % (goldens.get("cv2_version", "?"), "; ".join(report)))
    bad = [r for r in report if ": 0 of" not in r]
    assert not bad, ("this OpenCV runs neither the classic warpPerspective / remap kernels nor any candidate of the float32 family "
                     "(oracle.WARP_FAMILY / VARIANT_REMAP; the module fixture tried them all and kept the closest it was told to): " + "; ".join(bad))


def test_jpeg_codec_of_this_opencv(goldens, repo_rig):
    """Row f4 against cv2 itself: cv2.imread of the reference's camera files and the files cv2.imwrite writes (main.py:74-77, surroundBEV.py:340,
    Tools/undistort.py:73) against the JPEG oracle -- which is already pinned against Pillow's libjpeg-turbo (tests/test_jpeg_oracle.py); a
    difference here would mean that this OpenCV's bundled codec is configured differently."""
    from oracle import jpeg as JO
    from tests import _jpeg_common as JC

    cams = JC.repo_camera_jpegs()
    for n in GC.CAMS:
        check(goldens, "imread_" + n, JO.imdecode(cams[n]))
    im = GC.jpeg_test_image()
    check(goldens, "imwrite_default", np.frombuffer(JO.imencode(im), np.uint8))
    check(goldens, "imwrite_q100", np.frombuffer(JO.imencode(im, 100), np.uint8))
