"""Batch fisheye undistortion of a directory of images -- the reference's Tools/undistort.py (:25-77) on the GPU.

Same flags and defaults (Tools/undistort.py:7-23); the maps come from bevw_fisheye_remapper_create (k_fisheye_map) with the
optical-axis offsets of :45-46, the pixels from bevw_remap (all images of the directory in one batch).  File decode / encode
reads and writes .jpg files with the GPU codec (cameracalibration_amd/imgcodecs.py, bit-exact against libjpeg-turbo = cv2.imread / cv2.imwrite)
and uses Pillow for the other formats (same codec as cv2: libpng).
"""
from __future__ import annotations

import argparse
import ctypes as C
import os

import numpy as np

from .. import _ffi
from .._ffi import check, f64, lib, ptr

# the reference's command line (Tools/undistort.py:7-23): flag, default, type, what it is
_OPTIONS = (
    ("width", 1280, int, "frame width in pixels"),
    ("height", 1024, int, "frame height in pixels"),
    ("load", True, bool, "read K / D from -path_k / -path_d (otherwise the built-in calibration)"),
    ("path_read", "./data/", str, "directory of the distorted images"),
    ("path_save", "./", str, "directory for the undistorted images"),
    ("path_k", "./data/camera_0_K.npy", str, "camera matrix file (.npy)"),
    ("path_d", "./data/camera_0_D.npy", str, "fisheye distortion coefficients file (.npy)"),
    ("focalscale", 1, float, "focal length scale of the undistorted view"),
    ("sizescale", 1, float, "size scale of the undistorted view"),
    ("offset_h", 0, float, "horizontal shift of the optical axis in the undistorted view"),
    ("offset_v", 0, float, "vertical shift of the optical axis in the undistorted view"),
    ("srcformat", "jpg", str, "extension of the input files (jpg / png)"),
    ("dstformat", "jpg", str, "extension of the output files (jpg / png)"),
    ("quality", 100, int, "jpg quality 0-100, or png compression 9-0"),
    ("name", None, str, "output file name prefix"),
)
parser = argparse.ArgumentParser(description="Fisheye Camera Undistortion")
for _flag, _default, _type, _help in _OPTIONS:
    parser.add_argument("-" + _flag, default=_default, type=_type, help=_help)

# Tools/undistort.py:28-32: the built-in calibration used with -load False (identical to data/front's K and D)
DEFAULT_K = np.array([[350.4931893001142, 0.0, 647.6297467576265],
                      [0.0, 352.43072872484805, 513.5196785119657],
                      [0.0, 0.0, 1.0]])
DEFAULT_D = np.array([[-0.03367245449576437], [0.015380779195912842], [-0.018654590946883556], [0.0058128945633924185]])


class Undistorter:
    """The map set of one (K, D, size, scales, offsets) on the device."""

    def __init__(self, K, D, width, height, focalscale=1.0, sizescale=1.0, offset_h=0.0, offset_v=0.0, device=0):
        _ffi.require_device()
        self.width, self.height = int(width), int(height)
        r = C.c_void_p()
        check(lib().bevw_fisheye_remapper_create(device, self.width, self.height, ptr(f64(K, 9)), ptr(f64(D, 4)),
                                                 float(focalscale), float(sizescale), float(offset_h), float(offset_v),
                                                 C.byref(r)))
        self._r = r
        dims = np.zeros(4, np.int32)
        check(lib().bevw_remapper_dims(r, ptr(dims)))
        self.out_w, self.out_h = int(dims[2]), int(dims[3])

    def maps(self):
        m1 = np.empty((self.out_h, self.out_w, 2), np.int16)
        m2 = np.empty((self.out_h, self.out_w), np.uint16)
        check(lib().bevw_remapper_get_maps(self._r, ptr(m1), ptr(m2)))
        return m1, m2

    def __call__(self, images):
        """uint8 [B, height, width, 3] (or one [height, width, 3]) -> undistorted images of the map size."""
        imgs = np.ascontiguousarray(images)
        single = imgs.ndim == 3
        if single:
            imgs = imgs[np.newaxis]
        if imgs.dtype != np.uint8 or imgs.shape[1:] != (self.height, self.width, 3):
            raise Exception("images must be uint8 [B, {}, {}, 3]".format(self.height, self.width))
        out = np.empty((imgs.shape[0], self.out_h, self.out_w, 3), np.uint8)
        check(lib().bevw_remap(self._r, ptr(imgs), imgs.shape[0], ptr(out)))
        return out[0] if single else out

    def close(self):
        if self._r:
            lib().bevw_remapper_destroy(self._r)
            self._r = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def main(argv=None):
    from PIL import Image

    args = parser.parse_args(argv)
    if not args.load:
        camera_mat, dist_coeff = DEFAULT_K, DEFAULT_D
    else:
        if not os.path.exists(args.path_k):
            raise Exception("Camera K File Path not exist")
        if not os.path.exists(args.path_d):
            raise Exception("Camera D File Path not exist")
        camera_mat, dist_coeff = np.load(args.path_k), np.load(args.path_d)
    if not os.path.exists(args.path_read):
        raise Exception("Original Image Read Path not exist")
    if not os.path.exists(args.path_save):
        raise Exception("Undistortion Image Save Path not exist")
    und = Undistorter(camera_mat, dist_coeff, args.width, args.height, args.focalscale, args.sizescale, args.offset_h,
                      args.offset_v)
    names = [f for f in os.listdir(args.path_read) if f[-4:] == '.' + args.srcformat]
    # The reference streams image by image (Tools/undistort.py:60-77).  Here the directory is walked in bounded chunks:
    # host and device memory stay at CHUNK images whatever the directory holds, and a file of another size is reported
    # and skipped instead of aborting the run.
    CHUNK = 64
    index, done = 1, 0
    # .jpg in / .jpg out go through the GPU codec (imgcodecs: bit-exact against libjpeg-turbo, i.e. the bytes cv2.imread / cv2.imwrite give);
    # other formats, and JPEG flavours the codec refuses (progressive, CMYK ...), through Pillow as before.
    from .. import imgcodecs

    codec = imgcodecs.JpegCodec(und.device if hasattr(und, "device") else 0) if 'jpg' in (args.srcformat, args.dstformat) else None

    def read_chunk(files):
        """[(name, BGR array)] of the files that have the expected size."""
        raws = [(f, open(os.path.join(args.path_read, f), "rb").read()) for f in files]
        out, gpu = {}, []
        for f, raw in raws:
            ok = False
            if codec is not None and args.srcformat == 'jpg':
                try:
                    info = imgcodecs.probe(raw)
                    ok = True
                    if (info["height"], info["width"]) != (args.height, args.width):
                        print("{}: {}x{} is not {}x{}, skipped".format(f, info["width"], info["height"], args.width, args.height))
                        continue
                    gpu.append((f, raw, (info["components"], info["h_samp"], info["v_samp"], info["orientation"])))
                except _ffi.BevwError as e:
                    # not silent: this file is outside the GPU codec's subset and is decoded on the CPU by Pillow (as every file was before
                    # the codec existed); the remap itself still runs on the GPU
                    print("{}: decoded by Pillow, not by the GPU codec ({})".format(f, e))
                    ok = False
            if not ok:
                import io
                from PIL import ImageOps
                img = np.ascontiguousarray(np.asarray(ImageOps.exif_transpose(Image.open(io.BytesIO(raw))).convert("RGB"))[:, :, ::-1])   # (cv2.imread applies the EXIF orientation)
                if img.shape != (args.height, args.width, 3):
                    print("{}: {}x{} is not {}x{}, skipped".format(f, img.shape[1], img.shape[0], args.width, args.height))
                    continue
                out[f] = img
        for geom in sorted({g for _, _, g in gpu}):   # one geometry (components, sampling, EXIF orientation) per decode batch
            group = [(f, raw) for f, raw, g in gpu if g == geom]
            dec = codec.decode([raw for _, raw in group])
            for (f, _), img in zip(group, dec):
                out[f] = img
        return [(f, out[f]) for f in files if f in out]

    for c0 in range(0, len(names), CHUNK):
        pairs = read_chunk(names[c0:c0 + CHUNK])
        if not pairs:
            continue
        keep = [f for f, _ in pairs]
        out = und(np.stack([img for _, img in pairs]))
        jpgs = codec.encode(out, min(100, max(1, args.quality))) if (codec is not None and args.dstformat == 'jpg') else None
        for i, (filename, img) in enumerate(zip(keep, out)):
            print(filename)
            if args.name is not None:
                filename = args.name + '_{:04d}.'.format(index) + args.srcformat
                index += 1
            if jpgs is not None:
                # cv2.imwrite(..., [IMWRITE_JPEG_QUALITY, q]) keeps libjpeg's default 4:2:0 subsampling at every quality
                with open(os.path.join(args.path_save, filename[:-4] + '.jpg'), "wb") as fh:
                    fh.write(jpgs[i])
            else:
                pil = Image.fromarray(np.ascontiguousarray(img[:, :, ::-1]))
                if args.dstformat == 'png':
                    pil.save(os.path.join(args.path_save, filename[:-4] + '.png'), compress_level=min(9, max(0, args.quality)))
                else:
                    pil.save(filename[:-4] + '.' + args.dstformat)
            done += 1
    if codec is not None:
        codec.close()
    return done



if __name__ == '__main__':
    main()
