"""Parity soak (GPU box): random fisheye / pinhole remappers (scales, axis offsets), random homography warps, resize, translate.
Usage: python tools/soak_warps.py FIRST_SEED LAST_SEED   (prints every mismatch; BEVW_SOAK_SECONDS: stop there and report what was done.
Round 6: the script used to take N_CASES only and printed its result at the very end -- two calls that gave it a seed RANGE were cut off by
their time-out without a line; it now takes the range, honours the time budget and always reports.)"""
import os, sys, time, ctypes as C, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from cameracalibration_amd import _ffi, workloads as W
from oracle import oracle as O
O.build()
L = _ffi.lib(); bad = 0; done = 0
K0, D0, _ = W.repo_rig()["front"]
first, last = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (0, int(sys.argv[1]))
t_end = time.time() + float(os.environ.get("BEVW_SOAK_SECONDS", "1e9"))
for seed in range(first, last):
    if time.time() > t_end: break
    rng = np.random.default_rng(90000 + seed)
    fw = int(rng.choice([64, 97, 160, 200, 322, 400, 640])); fh = int(rng.choice([48, 65, 128, 150, 258, 480]))
    A = np.diag([fw / 1280.0, fh / 1024.0, 1.0]); K = A @ K0
    fs, ss = float(rng.choice([0.4, 0.5, 0.8, 1.0, 1.3])), float(rng.choice([0.5, 1.0, 1.5, 2.0]))
    oh, ov = float(rng.choice([0, -7.5, 12])), float(rng.choice([0, 3.25, -20]))
    batch = int(rng.choice([1, 2, 5, 9]))
    imgs = rng.integers(0, 256, (batch, fh, fw, 3), dtype=np.uint8)
    pin = bool(rng.integers(0, 2))
    r = C.c_void_p()
    if pin:
        nd = int(rng.choice([4, 5, 8])); D = (rng.standard_normal(nd) * np.array([0.2, 0.05, 0.002, 0.002, 0.01, 0.01, 0.01, 0.01])[:nd])
        _ffi.check(L.bevw_pinhole_remapper_create(0, fw, fh, _ffi.ptr(_ffi.f64(K, 9)), _ffi.ptr(np.ascontiguousarray(D, np.float64)), nd, fs, ss, oh, ov, C.byref(r)))
        Kd = O.camera_mat_dst(K, fw, fh, fs, ss, oh, ov); m1, m2 = O.init_undistort_rectify_map(K, D, Kd, (int(fw * ss), int(fh * ss)))
    else:
        D = D0.reshape(-1) * float(rng.choice([0, 1, 2]))
        _ffi.check(L.bevw_fisheye_remapper_create(0, fw, fh, _ffi.ptr(_ffi.f64(K, 9)), _ffi.ptr(_ffi.f64(D, 4)), fs, ss, oh, ov, C.byref(r)))
        Kd = O.camera_mat_dst(K, fw, fh, fs, ss, oh, ov); m1, m2 = O.fisheye_init_undistort_rectify_map(K, D, Kd, (int(fw * ss), int(fh * ss)))
    dims = np.zeros(4, np.int32); _ffi.check(L.bevw_remapper_dims(r, _ffi.ptr(dims)))
    g1 = np.empty((dims[3], dims[2], 2), np.int16); g2 = np.empty((dims[3], dims[2]), np.uint16); _ffi.check(L.bevw_remapper_get_maps(r, _ffi.ptr(g1), _ffi.ptr(g2)))
    out = np.empty((batch, dims[3], dims[2], 3), np.uint8); _ffi.check(L.bevw_remap(r, _ffi.ptr(imgs), batch, _ffi.ptr(out)))
    ok = np.array_equal(g1, m1) and np.array_equal(g2, m2) and all(np.array_equal(out[b], O.remap(imgs[b], m1, m2)) for b in range(batch))
    L.bevw_remapper_destroy(r)
    # warpPerspective with a random homography, resize, translate
    H = np.eye(3) + rng.standard_normal((3, 3)) * np.array([[0.2, 0.2, 20], [0.2, 0.2, 20], [5e-4, 5e-4, 0.0]])
    dw, dh = int(rng.choice([33, 64, 131, 250])), int(rng.choice([17, 64, 99, 200]))
    wout = np.empty((batch, dh, dw, 3), np.uint8); _ffi.check(L.bevw_warp_perspective_u8c3(0, _ffi.ptr(imgs), fw, fh, _ffi.ptr(_ffi.f64(H, 9)), dw, dh, batch, _ffi.ptr(wout)))
    ok = ok and all(np.array_equal(wout[b], O.warp_perspective(imgs[b], H, (dw, dh))) for b in range(batch))
    f = float(rng.choice([0.23, 0.5, 0.77, 1.0, 1.41, 2.0, 3.7])); ds = np.zeros(2, np.int32)
    if L.bevw_resize_dsize(fw, fh, f, f, _ffi.ptr(ds)) == 0 and ds[1] <= 65535:
        rout = np.empty((batch, ds[1], ds[0], 3), np.uint8); _ffi.check(L.bevw_resize_linear_u8c3(0, _ffi.ptr(imgs), fw, fh, f, f, batch, _ffi.ptr(rout)))
        ok = ok and all(np.array_equal(rout[b], O.resize_linear(imgs[b], f, f)) for b in range(batch))
    sx, sy = int(rng.integers(-fw, fw)), int(rng.integers(-fh, fh)); tout = np.empty_like(imgs)
    _ffi.check(L.bevw_translate_u8c3(0, _ffi.ptr(imgs), fw, fh, sx, sy, batch, _ffi.ptr(tout)))
    ok = ok and all(np.array_equal(tout[b], O.translate(imgs[b], sx, sy)) for b in range(batch))
    done += 1
    if not ok: bad += 1; print("MISMATCH seed", seed, fw, fh, fs, ss, pin, flush=True)
print("soak of remappers / warps / resize / translate, seeds", first, "..", "cases run", done, "mismatches", bad)
