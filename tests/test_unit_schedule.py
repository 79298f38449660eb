"""The unit schedule (cameracalibration_amd/csrc/bevw_unit.h) checked WITHOUT a GPU.

tests/native/unit_emulate.cpp is compiled with hipcc (only its host part runs): it runs the host-side plan compiler (unit_compile:
k-d partition of the BEV into units, group lists, LDS pair addresses) and a CPU emulation of the kernel body that uses the kernels' own
pair-conversion / dot-product helpers (their host versions restate v_perm_b32 / v_dot4_u32_u8 / v_dot2_u32_u16), and compares every
stored pixel with cv2.remap's fixed-point formula evaluated from the LUT.  Here it is driven with
  * its built-in synthetic rig (two cameras, seam, hole, blend weight, sparse corner, car sprite), and
  * the oracle's tables of the bench rigs (BASELINE config 3 direct / blend, the 4K rig, the undistort map): the pixels the units store
    must equal the oracle's BevGenerator output byte for byte, and the partition's request arithmetic must stay at the level DESIGN.md
    section 4 quotes."""
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest

from cameracalibration_amd import workloads as W
from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
pytestmark = pytest.mark.skipif(not (os.path.exists(HIPCC) or shutil.which("hipcc")), reason="hipcc not available")


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("unit") / "unit_emulate")
    from tests import _native_build

    _native_build.build(os.path.join(ROOT, "tests", "native", "unit_emulate.cpp"), out)
    return out


def test_units_on_synthetic_tables(exe):
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    # direct, the same tables as blend weights, direct with a car sprite; then wide plans (21-bit fractions, fp32 interpolation): direct, blend
    assert r.stdout.count("unit schedule ok") == 5


@pytest.mark.parametrize("pitch", [None, "aligned"])
def test_units_with_the_16_byte_store_format(tmp_path, pitch):
    """-DBEVW_UNIT_STORE16=2 (round 6's store-format A/B, off by default): the wave-store of whole lane quads as 3 x 16 bytes -- the funnel
    unit_repack16 the kernel executes with DPP moves -- in the host emulator: every claimed byte stored exactly once, every pixel right, on the
    synthetic tables and 25 random rigs, dense rows and rows of whole sectors (odd widths leave lane quads with a masked lane: the 12-byte path)."""
    from tests import _native_build

    exe = str(tmp_path / "unit_emulate_store16")
    _native_build.build(os.path.join(ROOT, "tests", "native", "unit_emulate.cpp"), exe, extra=["-DBEVW_UNIT_STORE16=2"])
    env = dict(os.environ)
    if pitch:
        env["BEVW_EMU_PITCH"] = pitch
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and r.stdout.count("unit schedule ok") == 5, r.stdout + r.stderr
    env["BEVW_EMU_FUZZ"] = "900 25"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def _run(exe, tmp_path, luts, masks, frames, car, fw, fh, bw, bh, blend=False, fracs=None):
    """luts: [(int16 [bh,bw,2], uint16 [bh,bw])], masks: [uint8 [bh,bw]], frames: uint8 [n, ncams, fh, fw, 3];
    fracs: [uint32 [bh,bw,2]] 21-bit fractions -> a wide plan (the analytic projection mode)"""
    inp, outp = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(inp, "wb") as f:
        f.write(struct.pack("<8i", fw, fh, bw, bh, len(luts), frames.shape[0], int(car is not None), int(blend) | (2 if fracs is not None else 0)))
        for i, ((m1, m2), mk) in enumerate(zip(luts, masks)):
            f.write(np.ascontiguousarray(m1, np.int16).tobytes())
            f.write(np.ascontiguousarray(m2, np.uint16).tobytes())
            f.write(np.ascontiguousarray(mk, np.uint8).tobytes())
            if fracs is not None:
                f.write(np.ascontiguousarray(fracs[i], np.uint32).tobytes())
        f.write(np.ascontiguousarray(frames, np.uint8).tobytes())
        if car is not None:
            f.write(np.ascontiguousarray(car, np.uint8).tobytes())
    r = subprocess.run([exe, inp, outp], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    raw = open(outp, "rb").read()
    nunits, claimed, lines, sectors = struct.unpack("<4i", raw[:16])
    written = np.frombuffer(raw, np.uint8, bw * bh, 16).reshape(bh, bw)
    img = np.frombuffer(raw, np.uint8, frames.shape[0] * bh * bw * 3, 16 + bw * bh).reshape(frames.shape[0], bh, bw, 3)
    return dict(units=nunits, claimed=claimed, lines=lines, sectors=sectors, written=written, img=img, log=r.stdout)


def _mask2d(m):
    return m[..., 0] if m.ndim == 3 else m


@pytest.mark.parametrize("name,cfg,rig,blend,max_requests", [
    # requests per frame = distinct source lines + write sectors of the partition; every base tile of config 3 is a unit tile; round 2 paid ~136 k
    ("config3_direct", W.CONFIG_S, W.rig_s, False, 112_000),
    ("config3_blend", W.CONFIG_S, W.rig_s, True, None),
    ("rig_4k_blend", W.CONFIG_4K, W.rig_4k, True, None),
])
def test_units_on_bench_rigs_match_the_oracle(exe, tmp_path, name, cfg, rig, blend, max_requests):
    O.build()
    gen = O.RefBevGenerator(rig(), cfg, blend=blend, balance=False)
    fw, fh, bw, bh = cfg["FRAME_WIDTH"], cfg["FRAME_HEIGHT"], cfg["BEV_WIDTH"], cfg["BEV_HEIGHT"]
    frames = W.synthetic_frames(1, fw, fh, kind="random")
    rng = np.random.default_rng(7)
    car = np.zeros((bh, bw, 3), np.uint8)
    cw, ch = cfg["CAR_WIDTH"], cfg["CAR_HEIGHT"]
    x0, y0 = (bw - cw) // 2 - 20, (bh - ch) // 2 - 20          # a sprite that overlaps the trapezoids by 20 pixels
    car[y0:y0 + ch + 40, x0:x0 + cw + 40] = rng.integers(0, 256, (ch + 40, cw + 40, 3), dtype=np.uint8)
    luts = [cam.bev_maps for cam in gen.cameras]
    masks = [_mask2d(m) for m in gen.masks]
    got = _run(exe, tmp_path, luts, masks, frames, car, fw, fh, bw, bh, blend)
    ref = gen(*[frames[0, i] for i in range(4)], car)
    w = got["written"] == 1
    assert w.sum() > 0.95 * bw * bh, got["log"]         # the units take all of the image but the base tiles with border footprints
    assert np.array_equal(got["img"][0][w], ref[w]), name
    assert not got["img"][0][~w].any()
    if max_requests is not None:
        assert got["lines"] + got["sectors"] <= max_requests, got["log"]
    print(got["log"].strip())


def test_units_on_the_undistort_map_match_the_oracle(exe, tmp_path):
    """Config 2: cv2.remap through the fisheye undistort maps = a one-camera plan with mask 255 everywhere."""
    O.build()
    K, D = W.undistort_calibration()
    cfg = W.CONFIG_UNDISTORT
    fw, fh = cfg["FRAME_WIDTH"], cfg["FRAME_HEIGHT"]
    Kd = O.camera_mat_dst(K, fw, fh, cfg["FOCAL_SCALE"], cfg["SIZE_SCALE"])
    m1, m2 = O.fisheye_init_undistort_rectify_map(K, D, Kd, (int(fw * cfg["SIZE_SCALE"]), int(fh * cfg["SIZE_SCALE"])))
    img = W.synthetic_frames(1, fw, fh, kind="random")[0, :1]
    got = _run(exe, tmp_path, [(m1, m2)], [np.full(m1.shape[:2], 255, np.uint8)], img[None], None, fw, fh, m1.shape[1], m1.shape[0])
    ref = O.remap(img[0], m1, m2)
    w = got["written"] == 1
    assert w.sum() > 0.9 * w.size, got["log"]
    assert np.array_equal(got["img"][0][w], ref[w])
    print(got["log"].strip())


@pytest.mark.parametrize("blend", [False, True])
def test_wide_units_from_the_analytic_projection_match_its_specification(exe, tmp_path, blend):
    """The analytic projection mode on the unit schedule (csrc/bevwarp.hip: analytic_units_build): positions from oracle/np_analytic.project
    (fp64), fractions rounded to 21 bits, fp32 interpolation from the LDS patch -- against the fp64 specification of the mode
    (np_analytic.AnalyticBevGenerator): never more than 1 LSB apart, >= 99.9 % of the bytes identical (the GPU test's bar, tests/test_analytic.py)."""
    from oracle import np_analytic

    O.build()
    cfg, rig = W.CONFIG_S, W.rig_s()
    fw, fh, bw, bh = cfg["FRAME_WIDTH"], cfg["FRAME_HEIGHT"], cfg["BEV_WIDTH"], cfg["BEV_HEIGHT"]
    spec = np_analytic.AnalyticBevGenerator(rig, cfg, blend=blend)
    luts, fracs = [], []
    for px, py, valid in spec.proj:
        sx, sy = np.floor(px), np.floor(py)
        m1 = np.stack([sx, sy], -1).astype(np.int16)
        m1[~valid] = -32768
        fr = np.stack([np.minimum(np.rint((px - sx) * 2.0 ** 21), 2 ** 21 - 1), np.minimum(np.rint((py - sy) * 2.0 ** 21), 2 ** 21 - 1)], -1).astype(np.uint32)
        fr[~valid] = 0
        luts.append((m1, np.zeros((bh, bw), np.uint16)))
        fracs.append(fr)
    masks = [_mask2d(m) for m in spec.ref.masks]
    frames = W.synthetic_frames(1, fw, fh, kind="random")
    rng = np.random.default_rng(11)
    car = np.zeros((bh, bw, 3), np.uint8)
    car[400:700, 420:660] = rng.integers(0, 256, (300, 240, 3), dtype=np.uint8)
    got = _run(exe, tmp_path, luts, masks, frames, car, fw, fh, bw, bh, blend, fracs)
    want = spec(*[frames[0, i] for i in range(4)], car)
    w = got["written"] == 1
    assert w.sum() > 0.95 * bw * bh, got["log"]
    d = np.abs(got["img"][0].astype(np.int32) - want.astype(np.int32))[w]
    assert d.max() <= 1, int(d.max())
    assert (d == 0).mean() >= 0.999, float((d == 0).mean())
    print(got["log"].strip(), "| identical: %.5f" % (d == 0).mean())


@pytest.mark.parametrize("seed,pitch", [(2, None), (5, None), (8, "aligned")])
def test_units_on_random_rigs(exe, seed, pitch):
    """Fuzz of the plan compiler + the emulated kernel body: random sizes, 1..4 cameras with smooth maps of random scale and orientation that leave
    the frame in places, band masks with overlaps and weight ramps, holes, wide plans with "no sample" patches, car sprites, dense rows or rows of
    whole sectors -- every stored pixel against the formula, every claimed quad stored exactly once (tests/native/unit_emulate.cpp, BEVW_EMU_FUZZ)."""
    env = dict(os.environ, BEVW_EMU_FUZZ="%d 25" % seed)
    if pitch:
        env["BEVW_EMU_PITCH"] = pitch
    r = subprocess.run([exe], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert r.stdout.count("unit schedule ok") == 25
