"""Row f4 on the GPU: the HIP JPEG decoder / encoder (through the C-ABI, cameracalibration_amd.imgcodecs) against the oracle
(oracle/jpegoracle.c) AND against Pillow's libjpeg-turbo, the library behind cv2.imread / cv2.imwrite (main.py:74-77,
surroundBEV.py:340).  Byte work: tolerance 0.  Run with `-m gpu` on an MI355X."""
import io

import numpy as np
import pytest

from cameracalibration_amd import workloads as W
from tests import _jpeg_common as JC

pytestmark = pytest.mark.gpu
pytest.importorskip("PIL")


@pytest.fixture(scope="module")
def IC():
    from cameracalibration_amd import _ffi, imgcodecs

    _ffi.require_device()
    return imgcodecs


@pytest.fixture(scope="module")
def codec(IC):
    c = IC.JpegCodec(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def JO():
    from oracle import jpeg

    jpeg.build()
    return jpeg


def test_decode_reference_camera_files(IC, codec, JO):
    files = list(JC.repo_camera_jpegs().values())
    got = codec.decode(files)
    info = codec.decode_info()
    assert info["images"] == 4 and (info["width"], info["height"]) == (1280, 1024) and info["rounds"] >= 1
    for i, raw in enumerate(files):
        want, planes = JO.imdecode(raw, planes=True)
        assert np.array_equal(got[i], want), f"file {i}: {np.count_nonzero(got[i] != want)} bytes differ from the oracle"
        assert np.array_equal(got[i], JC.pil_decode(raw)), f"file {i} differs from libjpeg-turbo"
        mine = codec.planes(i, sum(p.size for p in planes))   # the intermediate after the inverse DCT as well
        assert np.array_equal(mine, np.concatenate([p.reshape(-1) for p in planes]))


@pytest.mark.parametrize("sub,samp", JC.SUBSAMPLINGS)
def test_decode_matrix(codec, JO, sub, samp):
    for h, w in JC.SIZES + ((300, 420),):
        files = [JC.pil_encode(JC.image(h, w, kind), q, sub) for kind in (0, 1, 2) for q in (95, 30)]   # one geometry, six quantisers
        got = codec.decode(files)
        for i, f in enumerate(files):
            assert np.array_equal(got[i], JO.imdecode(f)), (h, w, i)
            assert np.array_equal(got[i], JC.pil_decode(f)), (h, w, i)


def test_decode_grey_restart_and_private_tables(codec, JO):
    im = JC.image(200, 300, 2)
    f = JC.pil_encode_gray(im[:, :, 0])
    assert np.array_equal(codec.decode([f])[0], JC.pil_decode(f))
    for kw in (dict(restart_marker_blocks=1), dict(restart_marker_blocks=3), dict(restart_marker_rows=1), dict(restart_marker_rows=4)):
        files = [JC.pil_encode(JC.image(200, 300, k), 90, 2, **kw) for k in (0, 1, 2)]
        got = codec.decode(files)
        for i, f in enumerate(files):
            assert np.array_equal(got[i], JC.pil_decode(f)), (kw, i)
    # optimize=True: every file carries its own Huffman tables -> several table sets inside one batch
    files = [JC.pil_encode(JC.image(120, 168, k, seed=s), 85, 2, optimize=True) for k in (0, 1, 2) for s in (0, 1)]
    got = codec.decode(files)
    assert codec.decode_info()["table_sets"] > 1
    for i, f in enumerate(files):
        assert np.array_equal(got[i], JO.imdecode(f)) and np.array_equal(got[i], JC.pil_decode(f)), i


def test_decode_larger_batch_many_rounds(codec):
    # 64 camera-sized files (the reference's four, re-encoded at several qualities): thousands of subsequences per image
    base = [JC.pil_decode(r) for r in JC.repo_camera_jpegs().values()]
    files = [JC.pil_encode(base[i % 4], 60 + (i % 8) * 5, 2) for i in range(32)]
    got = codec.decode(files)
    for i in (0, 5, 13, 31):
        assert np.array_equal(got[i], JC.pil_decode(files[i])), i
    assert codec.decode_info()["subsequences"] > 10000


def test_noise_at_quality_100_long_unsynchronised_runs(codec):
    """Noise at quality 100: blocks of ~1000 bits (longer than a subsequence), 40 ... 130 synchronisation rounds per image, many stretches
    of the in-block tail walked side by side (k_jpeg_sync: a wave that reaches another wave's stretch must ask for another round -- the soak
    of round 4 caught the version that looked at the successor's entry state first; profiles/r04/soak_jpeg_fail_seed4_case1613.npz is one of those cases)."""
    rng = np.random.default_rng(20260925)
    for (h, w, sub, n) in ((84, 542, 2, 5), (413, 256, 0, 4), (200, 333, 1, 3)):
        files = [JC.pil_encode(rng.integers(0, 256, (h, w, 3), dtype=np.uint8), 100, sub) for _ in range(n)]
        for _ in range(3):   # (the failure depended on which wave was first)
            got = codec.decode(files)
            for i, f in enumerate(files):
                assert np.array_equal(got[i], JC.pil_decode(f)), (h, w, sub, i)
        assert codec.decode_info()["rounds"] > 16


def test_4k_camera_files(codec):
    """BASELINE config 5's frames (3840 x 2160) and the reference's largest sample (ExtrinsicCalibration/data/img_src_back.jpg is 2560 x 2048):
    ~15 k subsequences per image, several per lane in the per-image kernels."""
    for h, w, q in ((2160, 3840, 90), (2048, 2560, 97)):
        files = [JC.pil_encode(JC.image(h, w, k), q, 2) for k in (2, 0)]
        got = codec.decode(files)
        for i, f in enumerate(files):
            assert np.array_equal(got[i], JC.pil_decode(f)), (h, w, i)
        assert codec.decode_info()["subsequences"] > 8000
    im = JC.image(2160, 3840, 2)
    assert codec.encode(im[None])[0] == JC.pil_encode(im)


def test_bench_sized_batch_decodes_into_frame_sets(IC, codec):
    """BASELINE's batch: 256 frame sets = 1024 camera files of 1280 x 960 in one call, written as the [256][4][960][1280][3] buffer
    bevw_run_device reads; sampled images against libjpeg-turbo, all of them through a checksum (identical files must give identical frames)."""
    from cameracalibration_amd import _ffi

    uniq = [JC.pil_encode(JC.image(960, 1280, 2, seed=s), 88, 2) for s in range(6)]
    files = [uniq[(i * 5 + i // 7) % 6] for i in range(1024)]
    frame = 960 * 1280 * 3
    d = _ffi.DeviceBuffer(1024 * frame)
    codec.decode_stage(files)
    codec.decode_run_device(d.ptr, frame, 1280 * 3)
    codec.sync()
    want = [JC.pil_decode(u) for u in uniq]
    sums = [int(w.astype(np.uint64).sum()) for w in want]
    for i0 in range(0, 1024, 64):
        got = d.download((64, 960, 1280, 3), offset=i0 * frame)
        for k in range(64):
            u = ((i0 + k) * 5 + (i0 + k) // 7) % 6
            assert int(got[k].astype(np.uint64).sum()) == sums[u], i0 + k
        assert np.array_equal(got[0], want[(i0 * 5 + i0 // 7) % 6]) and np.array_equal(got[63], want[((i0 + 63) * 5 + (i0 + 63) // 7) % 6])
    d.free()
    assert codec.decode_info()["images"] == 1024


@pytest.mark.parametrize("sub,samp", JC.SUBSAMPLINGS)
def test_encode_matrix(codec, JO, sub, samp):
    for h, w in JC.SIZES:
        for q in (95, 50, 100, 10):
            ims = np.stack([JC.image(h, w, kind) for kind in (0, 1, 2)])
            files = codec.encode(ims, q, samp)
            for k in range(3):
                assert files[k] == JO.imencode(ims[k], q, samp), (h, w, q, k)
                assert files[k] == JC.pil_encode(ims[k], q, sub), (h, w, q, k)


def test_encode_bev_sized_batch_and_pitched_rows(IC, codec):
    from cameracalibration_amd import _ffi

    ims = np.stack([JC.image(1080, 1080, k) for k in (2, 0, 1, 2)])
    files = codec.encode(ims)
    for k in range(4):
        assert files[k] == JC.pil_encode(ims[k]), k
    # rows of 1088 pixels (BEVW_PITCH_ALIGNED): the encoder reads the pitched device image directly
    pitched = np.zeros((2, 1080, 1088, 3), np.uint8)
    pitched[:, :, :1080] = ims[:2]
    pitched[:, :, 1080:] = 77   # padding columns must not leak into the file
    d = _ffi.DeviceBuffer(pitched.nbytes).upload(pitched)
    codec.encode_run_device(d.ptr, 2, 1080, 1080, pitched[0].nbytes, 1088 * 3)
    got = codec.files()
    d.free()
    assert got[0] == files[0] and got[1] == files[1]


def test_cv2_names_round_trip(IC, tmp_path):
    im = JC.image(97, 131, 2)
    p = str(tmp_path / "bev.jpg")
    assert IC.imwrite(p, im)
    raw = open(p, "rb").read()
    assert raw == JC.pil_encode(im)                      # cv2.imwrite defaults: quality 95, 4:2:0
    assert np.array_equal(IC.imread(p), JC.pil_decode(raw))
    ok, buf = IC.imencode(".jpg", im, [IC.IMWRITE_JPEG_QUALITY, 70])
    assert ok and buf.tobytes() == JC.pil_encode(im, 70)
    assert np.array_equal(IC.imdecode(buf), JC.pil_decode(buf.tobytes()))


def test_refusals_are_loud(IC, codec):
    from PIL import Image
    from cameracalibration_amd._ffi import BevwError

    im = JC.image(32, 32, 2)
    b = io.BytesIO()
    Image.fromarray(im).save(b, "JPEG", progressive=True)
    with pytest.raises(BevwError, match="progressive"):
        codec.decode([b.getvalue()])
    with pytest.raises(BevwError, match="not a JPEG"):
        codec.decode([b"\x89PNG\r\n\x1a\n" + bytes(64)])
    with pytest.raises(BevwError, match="one geometry"):
        codec.decode([JC.pil_encode(im), JC.pil_encode(JC.image(32, 40, 2))])
    with pytest.raises(BevwError):
        codec.decode([JC.pil_encode(im)[:300]])
    big = JC.pil_encode(JC.image(200, 300, 1), 95)
    with pytest.raises(BevwError, match="before their image is complete"):
        codec.decode([big, big[: len(big) * 2 // 3] + b"\xff\xd9"])     # the scan stops two thirds in: flagged, not silently grey
    assert np.array_equal(codec.decode([big])[0], JC.pil_decode(big))      # and the context is usable afterwards
    # garbage instead of entropy-coded data: no hang, no crash, nothing to compare
    rng = np.random.default_rng(5)
    head = big[: big.index(b"\xff\xda") + 14]
    junk = head + bytes(b if b != 255 else 254 for b in rng.integers(0, 256, 20000, dtype=np.uint8).tobytes()) + b"\xff\xd9"
    try:
        codec.decode([junk])
    except BevwError:
        pass
    assert np.array_equal(codec.decode([big])[0], JC.pil_decode(big))


def test_corrupted_entropy_data_never_hangs_or_crashes(IC, codec):
    """Random damage behind the scan header (bytes flipped, stretches zeroed, 0xFF / RSTn injected, tails cut): every call returns -- pixels or a
    BevwError -- and the context decodes a good file right afterwards."""
    from cameracalibration_amd._ffi import BevwError

    rng = np.random.default_rng(11)
    good = [JC.pil_encode(JC.image(240, 320, 2), 90, 2), JC.pil_encode(JC.image(240, 320, 1), 75, 2, restart_marker_blocks=7)]
    want = [JC.pil_decode(g) for g in good]
    outcomes = {"decoded": 0, "refused": 0}
    for trial in range(40):
        g = bytearray(good[trial % 2])
        sos = bytes(g).index(b"\xff\xda") + 14
        kind = trial % 5
        for _ in range(int(rng.integers(1, 12))):
            p = int(rng.integers(sos, len(g) - 2))
            if kind == 0: g[p] ^= 1 << int(rng.integers(0, 8))
            elif kind == 1: g[p:p + int(rng.integers(1, 400))] = bytes(int(rng.integers(1, 400)))
            elif kind == 2: g[p:p + 2] = bytes([0xFF, int(rng.choice([0xD0, 0xD3, 0xD7, 0x00, 0xFF]))])
            elif kind == 3: g = g[:p] + bytearray(b"\xff\xd9")
            else: g[p] = int(rng.integers(0, 256))
            if kind == 3: break
        try:
            out = codec.decode([bytes(g), good[trial % 2]])
            outcomes["decoded"] += 1
            assert np.array_equal(out[1], want[trial % 2])          # the undamaged file of the same batch is unaffected
        except BevwError:
            outcomes["refused"] += 1
        assert np.array_equal(codec.decode([good[trial % 2]])[0], want[trial % 2])
    assert outcomes["decoded"] + outcomes["refused"] == 40


def test_main_py_with_files_in_and_a_file_out(IC, JO, repo_rig, oracle):
    """main.py:72-84 (runBEV) + surroundBEV.py:340 end to end on compressed data: four camera FILES in, the stitched .jpg out, against
    cv2.imwrite(bev(*[cv2.imread(f) ...])) restated by the two oracles."""
    from cameracalibration_amd.SurroundBirdEyeView import surroundBEV as SB

    cams = JC.repo_camera_jpegs()
    files = [cams[n] for n in W.CAMERA_NAMES]
    cfg = dict(W.CONFIG_R, CAR_WIDTH=200, CAR_HEIGHT=350)
    ns = SB.BevGenerator.get_args()
    for k, v in cfg.items():
        setattr(ns, k, v)
    car = SB.padding(repo_rig.image("car"), cfg["BEV_WIDTH"], cfg["BEV_HEIGHT"])
    frames = [JO.imdecode(f) for f in files]
    for blend, balance, pitch in ((False, False, "dense"), (True, True, "dense"), (False, False, "aligned")):
        bev = SB.BevGenerator(blend=blend, balance=balance, rig=repo_rig.rig, output_pitch=pitch)
        ref = oracle.RefBevGenerator(repo_rig.rig, cfg, blend=blend, balance=balance)
        want = JO.imencode(ref(*frames, car))
        got = bev.jpeg([files, files[::-1]], car)
        assert len(got) == 2 and got[0] == want, (blend, balance, pitch)
        assert got[1] == JO.imencode(ref(*frames[::-1], car))


def test_empty_restart_segments_are_flagged(IC, codec):
    """a restart segment without a single entropy-coded byte (two adjacent RSTn markers, an RSTn right in front of EOI, a scan that
    is only EOI) owns no subsequence, so nothing downstream would look at its blocks -- such a file must be refused, never decoded from stale
    coefficients of an earlier batch."""
    from cameracalibration_amd._ffi import BevwError

    good = JC.pil_encode(JC.image(96, 128, 2), 90, 2, restart_marker_blocks=4)
    assert np.array_equal(codec.decode([good])[0], JC.pil_decode(good))      # (fills the coefficient buffer: the stale data a bad file would show)
    sos = good.index(b"\xff\xda") + 14
    body = good[sos:-2]
    first = body.index(b"\xff\xd0")
    second = body.index(b"\xff\xd1")
    # the segment between RST0 and RST1 removed: RST0 RST1 back to back (the marker count still matches DRI)
    adjacent = good[:sos] + body[:first + 2] + body[second:] + b"\xff\xd9"
    # the last segment removed: an RSTn right in front of EOI
    last = max(body.rfind(bytes([0xFF, 0xD0 + k])) for k in range(8))
    tail = good[:sos] + body[:last + 2] + b"\xff\xd9"
    plain = JC.pil_encode(JC.image(96, 128, 2), 90, 2)
    psos = plain.index(b"\xff\xda") + 14
    only_eoi = plain[:psos] + b"\xff\xd9"
    for bad in (adjacent, tail, only_eoi):
        with pytest.raises(BevwError):
            codec.decode([bad])
        with pytest.raises(BevwError):
            codec.decode([good, bad])
    assert np.array_equal(codec.decode([good])[0], JC.pil_decode(good))


def test_jpeg_pipeline_refuses_truncated_files_and_streams_identically(IC, JO, repo_rig, oracle):
    """BevGenerator.jpeg raises on a truncated camera file (it used to stitch undefined pixels), and jpeg_stream -- staging,
    kernels and fetch of consecutive batches overlapped, three codec contexts -- yields exactly what jpeg() returns batch by batch."""
    from cameracalibration_amd._ffi import BevwError
    from cameracalibration_amd.SurroundBirdEyeView import surroundBEV as SB

    cams = JC.repo_camera_jpegs()
    files = [cams[n] for n in W.CAMERA_NAMES]
    cfg = dict(W.CONFIG_R, CAR_WIDTH=200, CAR_HEIGHT=350)
    ns = SB.BevGenerator.get_args()
    for k, v in cfg.items():
        setattr(ns, k, v)
    car = SB.padding(repo_rig.image("car"), cfg["BEV_WIDTH"], cfg["BEV_HEIGHT"])
    bev = SB.BevGenerator(blend=True, balance=True, rig=repo_rig.rig)
    cut = files[2][: len(files[2]) * 3 // 5] + b"\xff\xd9"
    with pytest.raises(BevwError, match="before their image is complete"):
        bev.jpeg([files, [files[0], files[1], cut, files[3]]], car)
    # batches of different sizes and contents; the stream must reproduce the unpipelined calls in order
    perms = [files, files[::-1], [files[1], files[0], files[3], files[2]], [files[2], files[3], files[0], files[1]]]
    batches = [[perms[0], perms[1]], [perms[2]], [perms[3], perms[0], perms[1]], [perms[1]], [perms[2], perms[3]]]
    want = [bev.jpeg(b, car) for b in batches]
    got = list(bev.jpeg_stream(iter(batches), car))
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert g == w
    views = list(bev.jpeg_stream(batches[:2], car, copy=False))
    assert [[bytes(v) for v in b] for b in views] == want[:2]
    assert list(bev.jpeg_stream([], car)) == []
    ref = oracle.RefBevGenerator(repo_rig.rig, cfg, blend=True, balance=True)
    assert want[0][0] == JO.imencode(ref(*[JO.imdecode(f) for f in files], car))


def test_jpeg_stream_delivers_the_batch_in_flight_before_an_error_and_leaves_clean_slots(IC, JO, repo_rig, oracle):
    """when the staging of batch k + 1 fails (a file that is not a JPEG), batch k is on the GPU already and is good work -- the
    stream yields it, THEN raises; when the collect of batch k fails (a truncated file shows only after decoding), the generator raises and
    drains whatever is in flight.  Either way the next call on the same generator object starts from clean slots and is bit-exact."""
    from cameracalibration_amd._ffi import BevwError
    from cameracalibration_amd.SurroundBirdEyeView import surroundBEV as SB

    cams = JC.repo_camera_jpegs()
    files = [cams[n] for n in W.CAMERA_NAMES]
    cfg = dict(W.CONFIG_R, CAR_WIDTH=200, CAR_HEIGHT=350)
    ns = SB.BevGenerator.get_args()
    for k, v in cfg.items():
        setattr(ns, k, v)
    bev = SB.BevGenerator(blend=False, balance=False, rig=repo_rig.rig)
    good = bev.jpeg([files])
    not_a_jpeg = [files[0], b"definitely not a JPEG file", files[2], files[3]]
    got = []
    with pytest.raises(Exception):
        for out in bev.jpeg_stream([[files], [files[::-1]], [not_a_jpeg], [files]]):
            got.append(out)
    assert len(got) == 2 and got[0] == good and got[1] == bev.jpeg([files[::-1]])   # both batches before the bad one were delivered
    cut = files[2][: len(files[2]) * 3 // 5] + b"\xff\xd9"
    got = []
    with pytest.raises(BevwError, match="before their image is complete"):
        for out in bev.jpeg_stream([[files], [[files[0], files[1], cut, files[3]]], [files], [files]]):
            got.append(out)
    assert got == [good]
    assert list(bev.jpeg_stream([[files], [files]])) == [good, good] and bev.jpeg([files]) == good   # the slots are clean


def test_exif_orientations_are_applied_like_cv2_imread(IC, codec):
    """cv2.imread turns an image by its EXIF orientation tag (OpenCV's ExifTransform); so does the decoder, on the GPU.  Reference: the same
    file decoded by libjpeg-turbo (Pillow) and turned by Pillow's exif_transpose -- the EXIF specification's eight cases, odd sizes included."""
    from PIL import Image, ImageOps
    from cameracalibration_amd._ffi import BevwError

    for h, w in ((48, 64), (37, 53), (1, 7)):
        im = JC.image(h, w, 2)
        rgb = Image.fromarray(np.ascontiguousarray(im[:, :, ::-1]))
        files = {}
        for o in range(1, 9):
            exif = Image.Exif()
            exif[0x0112] = o
            b = io.BytesIO()
            rgb.save(b, "JPEG", quality=90, subsampling=2, exif=exif)
            files[o] = b.getvalue()
            want = np.ascontiguousarray(np.asarray(ImageOps.exif_transpose(Image.open(io.BytesIO(files[o]))).convert("RGB"))[:, :, ::-1])
            got = codec.decode([files[o], files[o]])
            assert got.shape[1:] == want.shape and np.array_equal(got[0], want) and np.array_equal(got[1], want), (h, w, o)
            assert np.array_equal(IC.imdecode(np.frombuffer(files[o], np.uint8)), want)
        with pytest.raises(BevwError, match="one orientation per batch"):
            codec.decode([files[1], files[6]])

