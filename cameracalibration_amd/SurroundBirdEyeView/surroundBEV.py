"""Host-side mirror of the reference's SurroundBirdEyeView/surroundBEV.py on top of libbevwarp (HIP, gfx950).

Same names, same argument meaning, same error behaviour as the reference module (file:line cited per symbol), so
``main.py``'s ``runBEV`` body (main.py:79-84) runs unchanged::

    from SurroundBirdEyeView import BevGenerator
    args = BevGenerator.get_args(); args.CAR_WIDTH = 200; args.CAR_HEIGHT = 350
    bev = BevGenerator(blend=True, balance=True)
    surround = bev(front, back, left, right)

Every pixel is produced on the GPU through the C-ABI (include/bevwarp.h); there is no cv2 and no NumPy compute path.
Additive API (not in the reference): ``BevGenerator(..., rig=..., device=..., schedule=...)``, ``bev.batch(frames)``,
``bev.run_device(...)``.
"""
from __future__ import annotations

import argparse
import ctypes as C
import os

import numpy as np

try:
    from .. import _ffi
except ImportError:
    # imported as a TOP-LEVEL package (sys.path points inside cameracalibration_amd/, the drop-in layout of main.py:5-7)
    import importlib
    import os as _os
    import sys as _sys

    _sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))))
    _ffi = importlib.import_module("cameracalibration_amd._ffi")
check, f64, lib, ptr = _ffi.check, _ffi.f64, _ffi.lib, _ffi.ptr

# surroundBEV.py:6-17 -- identical flags and defaults.  parse_known_args: importing this module must never abort
# because some other program's flags are on sys.argv (the reference's parse_args() does).
parser = argparse.ArgumentParser(description="Generate Surrounding Camera Bird Eye View")
parser.add_argument('-fw', '--FRAME_WIDTH', default=1280, type=int, help='Camera Frame Width')
parser.add_argument('-fh', '--FRAME_HEIGHT', default=1024, type=int, help='Camera Frame Height')
parser.add_argument('-bw', '--BEV_WIDTH', default=1000, type=int, help='BEV Frame Width')
parser.add_argument('-bh', '--BEV_HEIGHT', default=1000, type=int, help='BEV Frame Height')
parser.add_argument('-cw', '--CAR_WIDTH', default=250, type=int, help='Car Frame Width')
parser.add_argument('-ch', '--CAR_HEIGHT', default=400, type=int, help='Car Frame Height')
parser.add_argument('-fs', '--FOCAL_SCALE', default=1, type=float, help='Camera Undistort Focal Scale')
parser.add_argument('-ss', '--SIZE_SCALE', default=2, type=float, help='Camera Undistort Size Scale')
parser.add_argument('-blend', '--BLEND_FLAG', default=False, type=bool, help='Blend BEV Image (Ture/False)')
parser.add_argument('-balance', '--BALANCE_FLAG', default=False, type=bool, help='Balance BEV Image (Ture/False)')
args, _unknown = parser.parse_known_args(_ffi.own_argv(parser))

FRAME_WIDTH = args.FRAME_WIDTH
FRAME_HEIGHT = args.FRAME_HEIGHT
BEV_WIDTH = args.BEV_WIDTH
BEV_HEIGHT = args.BEV_HEIGHT
CAR_WIDTH = args.CAR_WIDTH
CAR_HEIGHT = args.CAR_HEIGHT
FOCAL_SCALE = args.FOCAL_SCALE
SIZE_SCALE = args.SIZE_SCALE

CAMERA_NAMES = ('front', 'back', 'left', 'right')


def _data_dir() -> str:
    """Where camera_<name>_{K,D,H}.npy live (surroundBEV.py:83-85 uses dirname(__file__) + '/data')."""
    return os.environ.get("BEVW_DATA_DIR") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")


def padding(img, width, height):
    """surroundBEV.py:28-41 -- centre the car sprite on a zero canvas (cv2.copyMakeBorder, constant 0).
    A one-off copy of a 250x400 sprite outside the per-frame path (surroundBEV.py:332-333)."""
    img = np.asarray(img)
    H, W = img.shape[0], img.shape[1]
    top = (height - H) // 2
    left = (width - W) // 2
    out = np.zeros((height, width) + img.shape[2:], dtype=img.dtype)
    out[top:top + H, left:left + W] = img
    return out


def color_balance(image, device: int = 0):
    """surroundBEV.py:43-55 on the GPU (k_channel_sums + k_gain)."""
    img = _ffi.as_u8_image(image)
    out = np.empty_like(img)
    check(lib().bevw_color_balance(device, ptr(img), 1, img.shape[1], img.shape[0], ptr(out)))
    return out


def luminance_balance(images, device: int = 0):
    """surroundBEV.py:57-79 on the GPU (k_vsum + k_lum_delta + k_lum_shift)."""
    imgs = [_ffi.as_u8_image(i) for i in images]
    if len(imgs) != 4 or any(i.shape != imgs[0].shape for i in imgs):
        raise Exception("luminance_balance expects [front, back, left, right] of one size")
    stack = np.stack(imgs)
    out = np.empty_like(stack)
    check(lib().bevw_luminance_balance(device, ptr(stack), 1, imgs[0].shape[1], imgs[0].shape[0], ptr(out)))
    return [out[i] for i in range(4)]


def _snapshot_config(blend=False, balance=False, device=0, schedule=_ffi.SCHED_AUTO) -> _ffi.bevw_config:
    return _ffi.bevw_config(int(FRAME_WIDTH), int(FRAME_HEIGHT), int(BEV_WIDTH), int(BEV_HEIGHT), int(CAR_WIDTH),
                            int(CAR_HEIGHT), float(FOCAL_SCALE), float(SIZE_SCALE), int(bool(blend)),
                            int(bool(balance)), int(device), int(schedule))


class _Engine:
    """Owns one bevw_handle (4 cameras + masks on the device)."""

    def __init__(self, rig, blend, balance, device, schedule, output_pitch=0):
        _ffi.require_device()
        self.cfg = _snapshot_config(blend, balance, device, schedule)
        h = C.c_void_p()
        check(lib().bevw_create(C.byref(self.cfg), C.byref(h)))
        self.h = h
        try:
            for i, (K, D, H) in enumerate(rig):
                check(lib().bevw_set_camera(self.h, i, ptr(f64(K, 9)), ptr(f64(D, 4)), ptr(f64(H, 9))))
            if output_pitch:
                check(lib().bevw_set_output_pitch(self.h, int(output_pitch)))
            check(lib().bevw_build(self.h))
        except Exception:
            self.close()
            raise
        self.und_size = (int(self.cfg.frame_width * self.cfg.size_scale), int(self.cfg.frame_height * self.cfg.size_scale))

    def close(self):
        if getattr(self, "h", None):
            lib().bevw_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Camera:
    """surroundBEV.py:81-117.  Inside a BevGenerator the four cameras share the generator's device tables; a
    stand-alone Camera(name) builds a private table set on first use."""

    def __init__(self, name, K=None, D=None, H=None):
        if name not in CAMERA_NAMES:
            raise Exception("name should be front/back/left/right")
        self.name = name
        if K is None:
            base = _data_dir()
            if os.path.exists(base + '/{}/camera_{}_K.npy'.format(name, name)):
                self.camera_mat = np.load(base + '/{}/camera_{}_K.npy'.format(name, name))
                self.dist_coeff = np.load(base + '/{}/camera_{}_D.npy'.format(name, name))
                self.homography = np.load(base + '/{}/camera_{}_H.npy'.format(name, name))
            else:
                # The reference ships its sample calibration as data/<name>/camera_<name>_{K,D,H}.npy; this package does
                # not redistribute the reference's files.  Without a data directory (or BEVW_DATA_DIR) the SAME twelve
                # matrices come from workloads.repo_rig() -- checked value for value against the reference's files by
                # tests/test_workloads.py -- so that BevGenerator() works out of the box as the reference's does.
                try:
                    from .. import workloads as _W
                except ImportError:
                    from cameracalibration_amd import workloads as _W
                self.camera_mat, self.dist_coeff, self.homography = _W.repo_rig()[name]
        else:
            self.camera_mat = np.array(K, dtype=np.float64).reshape(3, 3)
            self.dist_coeff = np.array(D, dtype=np.float64).reshape(-1, 1)
            self.homography = np.array(H, dtype=np.float64).reshape(3, 3)
        self.camera_mat_dst = self.get_camera_mat_dst()
        self._engine = None
        self._index = 0
        self._device = 0

    def _attach(self, engine, index):
        self._engine, self._index = engine, index

    def _eng(self) -> _Engine:
        if self._engine is None:
            trip = (self.camera_mat, self.dist_coeff, self.homography)
            self._engine = _Engine([trip] * 4, False, False, self._device, _ffi.SCHED_PER_PIXEL)
            self._index = 0
        return self._engine

    def get_camera_mat_dst(self):
        camera_mat_dst = self.camera_mat.copy()
        camera_mat_dst[0][0] *= FOCAL_SCALE
        camera_mat_dst[1][1] *= FOCAL_SCALE
        camera_mat_dst[0][2] = FRAME_WIDTH / 2 * SIZE_SCALE
        camera_mat_dst[1][2] = FRAME_HEIGHT / 2 * SIZE_SCALE
        return camera_mat_dst

    def get_undistort_maps(self):
        e = self._eng()
        w, h = e.und_size
        m1, m2 = np.empty((h, w, 2), np.int16), np.empty((h, w), np.uint16)
        check(lib().bevw_get_undistort_map(e.h, self._index, ptr(m1), ptr(m2)))
        return (m1, m2)

    def get_bev_maps(self):
        e = self._eng()
        w, h = e.cfg.bev_width, e.cfg.bev_height
        m1, m2 = np.empty((h, w, 2), np.int16), np.empty((h, w), np.uint16)
        check(lib().bevw_get_lut(e.h, self._index, ptr(m1), ptr(m2)))
        return (m1, m2)

    @property
    def undistort_maps(self):
        return self.get_undistort_maps()

    @property
    def bev_maps(self):
        return self.get_bev_maps()

    def undistort(self, img):
        e = self._eng()
        img = _ffi.as_u8_image(img)
        if img.shape[:2] != (e.cfg.frame_height, e.cfg.frame_width):
            raise Exception("image is {}x{}, FRAME is {}x{}".format(img.shape[1], img.shape[0], e.cfg.frame_width, e.cfg.frame_height))
        out = np.empty((e.und_size[1], e.und_size[0], 3), np.uint8)
        check(lib().bevw_camera_undistort(e.h, self._index, ptr(img), 1, ptr(out)))
        return out

    def warp_homography(self, img):
        e = self._eng()
        img = _ffi.as_u8_image(img)
        out = np.empty((e.cfg.bev_height, e.cfg.bev_width, 3), np.uint8)
        check(lib().bevw_camera_warp_homography(e.h, self._index, ptr(img), img.shape[1], img.shape[0], 1, ptr(out)))
        return out

    def raw2bev(self, img):
        e = self._eng()
        img = _ffi.as_u8_image(img)
        if img.shape[:2] != (e.cfg.frame_height, e.cfg.frame_width):
            raise Exception("image is {}x{}, FRAME is {}x{}".format(img.shape[1], img.shape[0], e.cfg.frame_width, e.cfg.frame_height))
        out = np.empty((e.cfg.bev_height, e.cfg.bev_width, 3), np.uint8)
        check(lib().bevw_camera_raw2bev(e.h, self._index, ptr(img), 1, ptr(out)))
        return out


class Mask:
    """surroundBEV.py:119-162.  `.mask` is the uint8 fillPoly mask, rasterised on the GPU (k_poly_outline/k_poly_fill).
    In BevGenerator.__call__ the mask is applied inside the fused stitch kernel, not by this object."""
    _blend = False

    def __init__(self, name, _engine=None, _index=None):
        if name not in CAMERA_NAMES:
            raise Exception("name should be front/back/left/right")
        self.name = name
        self._engine, self._index = _engine, (CAMERA_NAMES.index(name) if _index is None else _index)
        self._mask = None

    def _eng(self):
        if self._engine is None:
            ident = (np.eye(3), np.zeros(4), np.eye(3))
            self._engine = _Engine([ident] * 4, self._blend, False, 0, _ffi.SCHED_PER_PIXEL)
        return self._engine

    def get_mask(self, name=None):
        e = self._eng()
        idx = self._index if name is None else CAMERA_NAMES.index(name)
        m = np.empty((e.cfg.bev_height, e.cfg.bev_width), np.uint8)
        check(lib().bevw_get_mask(e.h, idx, ptr(m)))
        return m

    @property
    def mask(self):
        if self._mask is None:
            self._mask = self.get_mask()
        return self._mask

    def __call__(self, img):
        """surroundBEV.py:161-162 (Mask) / :279-280 (BlendMask) as a stand-alone GPU operation."""
        e = self._eng()
        img = _ffi.as_u8_image(img)
        if img.shape[:2] != (e.cfg.bev_height, e.cfg.bev_width):
            raise Exception("image is {}x{}, BEV is {}x{}".format(img.shape[1], img.shape[0], e.cfg.bev_width, e.cfg.bev_height))
        out = np.empty_like(img)
        check(lib().bevw_apply_mask(e.h, self._index, ptr(img), 1, ptr(out)))
        return out


class BlendMask(Mask):
    """surroundBEV.py:164-280.  `.mask` holds the uint8 alpha codes, `.weight` = float32(mask / 255.0) x 3 channels."""
    _blend = True

    @property
    def weight(self):
        return (np.repeat(self.mask[:, :, np.newaxis], 3, axis=2) / 255.0).astype(np.float32)


class BevGenerator:
    """surroundBEV.py:282-325.

    BevGenerator(blend, balance) reads the module-level `args` (get_args()) at construction exactly like the
    reference's init_args(), loads the four cameras' K/D/H and builds every table on the GPU.
    """

    def __init__(self, blend=args.BLEND_FLAG, balance=args.BALANCE_FLAG, *, rig=None, device=0,
                 schedule=_ffi.SCHED_AUTO, projection='lut', output_pitch='auto'):
        """blend / balance: as in the reference (surroundBEV.py:283).  Additive keywords: rig ({name: (K, D, H)} instead of the
        data directory), device, schedule, and projection -- 'lut' (default: the reference's table-driven path, bit-exact against
        the oracle) 'analytic' (inverse homography + fisheye model evaluated per frame and pixel in fp64, no tables; not the
        reference's fixed-point arithmetic -- see bevw_set_projection in include/bevwarp.h) or 'analytic_f32' (the same in fp32);
        output_pitch -- the row pitch of the DEVICE-side BEV images: 'auto' (default: 'aligned' wherever the tile plan with the table
        projection serves the handle, 'dense' otherwise), 'aligned' (rows of whole 64-byte sectors, cv::cuda::GpuMat style: see
        bevw_set_output_pitch in include/bevwarp.h), 'dense' (the reference's host layout) or a number of pixels.  Arrays returned to
        the host are dense either way; only run_device() callers see the pitch (``out_pitch`` pixels per row of their output buffer)."""
        self.init_args()
        if rig is None:
            self.cameras = [Camera('front'), Camera('back'), Camera('left'), Camera('right')]
        else:
            self.cameras = [Camera(n, *rig[n]) for n in CAMERA_NAMES]
        self.blend = blend
        self.balance = balance
        self.device = device
        auto = output_pitch == 'auto'
        if auto:   # the pitched layout needs the tile plan and the table projection (bevw_run_device refuses it otherwise)
            output_pitch = 'aligned' if (projection == 'lut' and schedule != _ffi.SCHED_PER_PIXEL) else 'dense'
        pitch = {'dense': _ffi.PITCH_DENSE, 'aligned': _ffi.PITCH_ALIGNED}.get(output_pitch, output_pitch)
        if not isinstance(pitch, int):
            raise Exception("output_pitch should be auto/dense/aligned or a number of pixels")
        rig_kdh = [(c.camera_mat, c.dist_coeff, c.homography) for c in self.cameras]
        try:
            self._engine = _Engine(rig_kdh, blend, balance, device, schedule, pitch)
        except _ffi.BevwError:
            if not (auto and pitch != _ffi.PITCH_DENSE):
                raise
            self._engine = _Engine(rig_kdh, blend, balance, device, schedule, _ffi.PITCH_DENSE)   # a rig the tile plan cannot serve
        self.out_pitch = int(lib().bevw_output_pitch(self._engine.h))   # pixels per row of run_device()'s output images
        modes = {'lut': _ffi.PROJ_LUT, 'analytic': _ffi.PROJ_ANALYTIC, 'analytic_f32': _ffi.PROJ_ANALYTIC_F32}
        if projection not in modes:
            raise Exception("projection should be lut/analytic/analytic_f32")
        self.projection = projection
        if projection != 'lut':
            check(lib().bevw_set_projection(self._engine.h, modes[projection]))
        for i, cam in enumerate(self.cameras):
            cam._attach(self._engine, i)
        cls = BlendMask if self.blend else Mask
        self.masks = [cls(n, self._engine, i) for i, n in enumerate(CAMERA_NAMES)]

    @staticmethod
    def get_args():
        return args

    def init_args(self):
        global FRAME_WIDTH, FRAME_HEIGHT, BEV_WIDTH, BEV_HEIGHT
        global CAR_WIDTH, CAR_HEIGHT, FOCAL_SCALE, SIZE_SCALE
        FRAME_WIDTH = args.FRAME_WIDTH
        FRAME_HEIGHT = args.FRAME_HEIGHT
        BEV_WIDTH = args.BEV_WIDTH
        BEV_HEIGHT = args.BEV_HEIGHT
        CAR_WIDTH = args.CAR_WIDTH
        CAR_HEIGHT = args.CAR_HEIGHT
        FOCAL_SCALE = args.FOCAL_SCALE
        SIZE_SCALE = args.SIZE_SCALE

    # ---- reference call ------------------------------------------------------------------------------------
    def __call__(self, front, back, left, right, car=None):
        c = self._engine.cfg
        images = [_ffi.as_u8_image(i, "camera frame") for i in (front, back, left, right)]
        for img in images:
            if img.shape[:2] != (c.frame_height, c.frame_width):
                raise Exception("camera frame is {}x{}, FRAME is {}x{}".format(img.shape[1], img.shape[0],
                                                                               c.frame_width, c.frame_height))
        car_p = None
        if car is not None:
            car = _ffi.as_u8_image(car, "car")
            if car.shape[:2] != (c.bev_height, c.bev_width):
                raise Exception("car must be padded to the BEV size (padding())")
            car_p = ptr(car)
        out = np.empty((c.bev_height, c.bev_width, 3), np.uint8)
        check(lib().bevw_run_cameras(self._engine.h, ptr(images[0]), ptr(images[1]), ptr(images[2]), ptr(images[3]), car_p,
                                     ptr(out)))
        return out

    # ---- additive: batches ---------------------------------------------------------------------------------
    def batch(self, frames, car=None):
        """frames uint8 [B, 4, FH, FW, 3] (front, back, left, right) -> uint8 [B, BH, BW, 3]."""
        c = self._engine.cfg
        frames = np.ascontiguousarray(frames)
        if frames.dtype != np.uint8 or frames.ndim != 5 or frames.shape[1:] != (4, c.frame_height, c.frame_width, 3):
            raise Exception("frames must be uint8 [B, 4, {}, {}, 3]".format(c.frame_height, c.frame_width))
        car_p = None
        if car is not None:
            car = _ffi.as_u8_image(car, "car")
            if car.shape[:2] != (c.bev_height, c.bev_width):
                raise Exception("car must be padded to the BEV size (padding())")
            car_p = ptr(car)
        out = np.empty((frames.shape[0], c.bev_height, c.bev_width, 3), np.uint8)
        check(lib().bevw_run(self._engine.h, ptr(frames), frames.shape[0], car_p, ptr(out)))
        return out

    # ---- additive: compressed in, compressed out (row f4) ----------------------------------------------------
    def _imgcodecs(self):
        try:
            from .. import imgcodecs
        except ImportError:   # top-level import of this package (main.py's drop-in layout)
            import importlib
            imgcodecs = importlib.import_module("cameracalibration_amd.imgcodecs")
        return imgcodecs

    class _JpegSlot:
        """One in-flight batch of the compressed path: a codec context (its own streams, staging memory and scratch) + the frame-set and BEV
        buffers of that batch."""

        def __init__(self, gen, imgcodecs):
            self.codec = imgcodecs.JpegCodec(gen.device)
            self.bufs = None
            self.batch = 0

        def reserve(self, gen, need):
            if self.bufs is None or self.bufs[0].nbytes < need[0] or self.bufs[1].nbytes < need[1]:
                if self.bufs is not None:
                    self.bufs[0].free()
                    self.bufs[1].free()
                self.bufs = (_ffi.DeviceBuffer(need[0], gen.device), _ffi.DeviceBuffer(need[1], gen.device))

    def _jpeg_sets(self, files):
        sets = [tuple(bytes(f) for f in s) for s in files]
        if not sets or any(len(s) != 4 for s in sets):
            raise Exception("files must be a non-empty sequence of (front, back, left, right) JPEG files")
        return sets

    def _jpeg_car(self, car):
        if car is None:
            return None
        c = self._engine.cfg
        car = _ffi.as_u8_image(car, "car")
        if car.shape[:2] != (c.bev_height, c.bev_width):
            raise Exception("car must be padded to the BEV size (padding())")
        if getattr(self, "_jpeg_car_buf", None) is None:
            self._jpeg_car_buf = _ffi.DeviceBuffer(car.nbytes, self.device)
        self._jpeg_car_buf.upload(car)
        return self._jpeg_car_buf.ptr

    def _jpeg_stage(self, slot, sets):
        """Host side of a batch: header parsing, the entropy-coded bytes into pinned memory, their upload enqueued (releases the GIL)."""
        c = self._engine.cfg
        info = slot.codec.decode_stage([f for s in sets for f in s])
        if (info["width"], info["height"]) != (c.frame_width, c.frame_height):
            raise Exception("camera files are {}x{}, FRAME is {}x{}".format(info["width"], info["height"], c.frame_width,
                                                                             c.frame_height))
        slot.batch = len(sets)

    def _jpeg_enqueue(self, slot, d_car, quality):
        """decode -> stitch -> encode of the staged batch as ONE asynchronous chain: the codec's stream and the engine's stream are ordered
        by events (bevw_wait_jpeg / bevw_jpeg_wait_engine), the host does not wait in between."""
        c = self._engine.cfg
        B = slot.batch
        frame = c.frame_height * c.frame_width * 3
        image = c.bev_height * self.out_pitch * 3
        slot.reserve(self, (B * 4 * frame, B * image))
        slot.codec.decode_run_device(slot.bufs[0].ptr, frame, c.frame_width * 3)
        slot.codec.engine_waits(self._engine.h)
        self.run_device(slot.bufs[0].ptr, B, d_car, slot.bufs[1].ptr, out_bytes=B * image)
        slot.codec.wait_engine(self._engine.h)
        slot.codec.encode_run_device(slot.bufs[1].ptr, B, c.bev_width, c.bev_height, image, self.out_pitch * 3, quality)

    def _jpeg_collect(self, slot, copy=True):
        """Wait for the batch, refuse truncated / corrupt camera files (their pixels are undefined), fetch the files."""
        files = slot.codec.files(copy=copy)   # synchronises with the codec's stream, i.e. with the whole chain
        short = slot.codec.decode_info()["short_images"]
        if short:
            raise _ffi.BevwError("{} of the {} camera files end before their image is complete (truncated / corrupt entropy-coded "
                                 "data)".format(short, 4 * slot.batch))
        return files

    def jpeg(self, files, car=None, quality=95):
        """main.py:74-84 + surroundBEV.py:340 with the pixels resident in HBM: ``files`` is a sequence of frame sets, each the four camera FILES' bytes
        (front, back, left, right: what main.py hands to cv2.imread); the result is one complete ``.jpg`` file per set -- the bytes
        cv2.imwrite(path, bev(front, back, left, right, car)) would write (libjpeg at quality 95, 4:2:0).  Decode, stitch and
        encode all run on the GPU (imgcodecs.JpegCodec); only compressed bytes cross PCIe.  Truncated or corrupt camera files raise."""
        sets = self._jpeg_sets(files)
        if getattr(self, "_jpeg_slots", None) is None:
            self._jpeg_slots = [BevGenerator._JpegSlot(self, self._imgcodecs())]
        slot = self._jpeg_slots[0]
        d_car = self._jpeg_car(car)
        self._jpeg_stage(slot, sets)
        self._jpeg_enqueue(slot, d_car, quality)
        return self._jpeg_collect(slot)

    def jpeg_stream(self, batches, car=None, quality=95, copy=True):
        """``jpeg()`` over an iterable of batches (each a sequence of frame sets), pipelined: while the GPU runs batch i (decode, stitch and
        encode chained by events on their streams), a host thread parses and stages batch i + 1 (its upload overlaps the kernels) and this
        thread fetches the files of batch i - 1.  Three codec contexts rotate, so the three stages never share a buffer.  Yields one list of
        files per batch, in order; results are identical to ``jpeg(batch)``.  ``copy=False`` yields ``memoryview`` slices of one host buffer
        per batch instead of ``bytes``.

        Throughput note: the pipeline keeps seven HIP streams busy and the runtime multiplexes them onto 4 hardware queues by default; the
        published figure is with 8 -- call ``cameracalibration_amd._ffi.prefer_hw_queues()`` before the first use of
        the package in the process (importing the package does not change the environment)."""
        from concurrent.futures import ThreadPoolExecutor

        imgcodecs = self._imgcodecs()
        if getattr(self, "_jpeg_slots", None) is None:
            self._jpeg_slots = []
        while len(self._jpeg_slots) < 3:
            self._jpeg_slots.append(BevGenerator._JpegSlot(self, imgcodecs))
        slots = self._jpeg_slots
        d_car = self._jpeg_car(car)
        it = iter(batches)

        def stage(k, batch):
            sets = self._jpeg_sets(batch)
            self._jpeg_stage(slots[k % 3], sets)
            return k

        with ThreadPoolExecutor(1) as pool:
            try:
                nxt = pool.submit(stage, 0, next(it))
            except StopIteration:
                return
            k, inflight = 0, None
            try:
                while nxt is not None:
                    try:
                        nxt.result()                              # batch k is staged (its upload is enqueued on its codec's stream)
                    except BaseException:
                        nxt = None
                        if inflight is not None:                  # batch k - 1 runs on the GPU and is good work: deliver it, then fail
                            done, inflight = inflight, None
                            yield self._jpeg_collect(slots[done % 3], copy)
                        raise
                    try:
                        nxt = pool.submit(stage, k + 1, next(it))  # the host side of batch k + 1 runs beside everything below
                    except StopIteration:
                        nxt = None
                    self._jpeg_enqueue(slots[k % 3], d_car, quality)
                    done, inflight = inflight, k
                    k += 1
                    if done is not None:
                        yield self._jpeg_collect(slots[done % 3], copy)   # batch k - 1, while the GPU runs batch k
                if inflight is not None:
                    done, inflight = inflight, None
                    yield self._jpeg_collect(slots[done % 3], copy)
            finally:
                # leaving early (an exception above, or the consumer dropped the generator): nothing may stay in flight on a slot the next
                # jpeg() / jpeg_stream() call re-uses -- wait for the staging thread, then for every codec stream
                if nxt is not None:
                    try:
                        nxt.result()
                    except BaseException:
                        pass
                if inflight is not None or nxt is not None:
                    for sl in slots:
                        try:
                            sl.codec.sync()
                        except Exception:
                            pass

    @property
    def out_image_bytes(self) -> int:
        """Bytes of ONE device-side BEV image as run_device() writes it: rows of ``out_pitch`` pixels (padding columns included)."""
        return self.out_pitch * self._engine.cfg.bev_height * 3

    def run_device(self, d_frames: int, batch: int, d_car, d_out: int, out_bytes: int = None) -> None:
        """Asynchronous launch on device-resident buffers (raw pointers from DeviceBuffer).

        out_bytes: the size of the buffer behind ``d_out``.  The library sees raw pointers and cannot check it, so a handle whose
        device images are pitched (``out_pitch != BEV_WIDTH``: the 'auto' / 'aligned' layouts) REQUIRES it -- a caller that sized its
        buffer for dense images (batch * BEV_HEIGHT * BEV_WIDTH * 3) would otherwise be overrun silently.  Size buffers with
        ``batch * bev.out_image_bytes`` or construct with ``output_pitch='dense'``."""
        need = int(batch) * self.out_image_bytes
        if out_bytes is None:
            if self.out_pitch != self._engine.cfg.bev_width:
                raise Exception("this BevGenerator writes device images with rows of {} pixels (output_pitch); pass out_bytes= (>= {} for "
                                "this batch) or construct it with output_pitch='dense'".format(self.out_pitch, need))
        elif int(out_bytes) < need:
            raise Exception("output buffer of {} bytes, {} images of {} bytes need {}".format(int(out_bytes), int(batch), self.out_image_bytes, need))
        check(lib().bevw_run_device(self._engine.h, d_frames, batch, d_car, d_out))

    def sync(self) -> None:
        check(lib().bevw_sync(self._engine.h))

    def timer_start(self) -> None:
        check(lib().bevw_timer_start(self._engine.h))

    def timer_stop(self) -> float:
        ms = C.c_float()
        check(lib().bevw_timer_stop(self._engine.h, C.byref(ms)))
        return float(ms.value)

    def timer_mark(self, slot: int) -> None:
        check(lib().bevw_timer_mark(self._engine.h, int(slot)))

    def timer_between(self, a: int, b: int) -> float:
        ms = C.c_float()
        check(lib().bevw_timer_between(self._engine.h, int(a), int(b), C.byref(ms)))
        return float(ms.value)

    def plan_info(self) -> dict:
        info = np.zeros(8, np.int32)
        check(lib().bevw_plan_info(self._engine.h, ptr(info)))
        return {"max_contributors": int(info[0]), "plan_usable": bool(info[1]), "schedule": int(info[2]),
                "tiles_x": int(info[3]), "tiles_y": int(info[4]), "tiles_staged": int(info[5]),
                "tiles_gather": int(info[6]), "tiles_border": int(info[7])}


def main():
    """surroundBEV.py:327-345 without the GUI: reads ./data like the reference and writes ./surround.jpg (:340).  The four camera .jpg files
    are decoded and the result is encoded on the GPU (imgcodecs: the bytes cv2.imread gives, the file cv2.imwrite writes); car.jpg is a PNG
    in the reference's data and goes through Pillow."""
    from PIL import Image

    try:
        from .. import imgcodecs
    except ImportError:
        import importlib
        imgcodecs = importlib.import_module("cameracalibration_amd.imgcodecs")

    base = _data_dir()
    front, back, left, right = (imgcodecs.imread(base + '/{0}/{0}.jpg'.format(n)) for n in CAMERA_NAMES)
    car = padding(np.ascontiguousarray(np.asarray(Image.open(base + '/car.jpg').convert("RGB"))[:, :, ::-1]), BEV_WIDTH, BEV_HEIGHT)
    bev = BevGenerator()
    surround = bev(front, back, left, right, car)
    imgcodecs.imwrite('./surround.jpg', surround)


if __name__ == '__main__':
    main()
