"""Decode the files a soak kept (gpurun_out/soak_jpeg_fail_*.npz) and compare with libjpeg-turbo (Pillow):  python tools/jpeg_repro.py [files.npz ...]"""
import glob, io, os, sys
import numpy as np
from PIL import Image
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cameracalibration_amd import imgcodecs
codec = imgcodecs.JpegCodec(0)
for path in (sys.argv[1:] or sorted(glob.glob("profiles/r04/soak_jpeg_fail_*.npz") + glob.glob("gpurun_out/soak_jpeg_fail_*.npz"))):
    z = np.load(path)
    files = [z[k].tobytes() for k in sorted(z.files, key=lambda s: int(s[1:]))]
    try:
        got = codec.decode(files)
    except Exception as e:
        print(os.path.basename(path), "REFUSED", repr(e)[:120]); continue
    bad = []
    for i, f in enumerate(files):
        ref = np.asarray(Image.open(io.BytesIO(f)).convert("RGB"))[:, :, ::-1]
        if not np.array_equal(got[i], ref):
            d = np.argwhere((got[i] != ref).any(axis=2))
            bad.append((i, len(d), tuple(d[0]), tuple(d[-1])))
    info = codec.decode_info()
    print(os.path.basename(path), "lib", os.environ.get("BEVW_LIB_PATH", "shipped").split("_")[-1], "rounds", info["rounds"], "differs:", bad or "nothing")
