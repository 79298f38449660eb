"""TEST INFRASTRUCTURE (never imported by the product): NumPy fp64 statement of the ANALYTIC projection mode
(bevw_set_projection(BEVW_PROJ_ANALYTIC)).

The reference has no such path -- its per-frame work is a table-driven cv2.remap (surroundBEV.py:116-117) -- so this is not
a restatement of reference code but the specification of the mode, written with the reference's own formulas:

    BEV pixel (x, y)  --H^-1-->  undistorted pixel (u, v)            ExCalibrator / warpPerspective geometry, surroundBEV.py:113-114
    (u, v)  --K'^-1, fisheye model (theta_d = theta (1 + k1 theta^2 + ...)), K-->  raw position (px, py)
                                                                     cv2.fisheye.initUndistortRectifyMap, surroundBEV.py:99-102
    bilinear interpolation of the 4 texels in fp64, round half to even, BORDER_CONSTANT 0 per tap
    zero where (u, v) leaves the undistorted image (warp_homography of an image is 0 there)

then the reference's own mask / blend weight / saturating sums / car.  Balance is not restated here (the GPU tests cover
blend on / off; balance goes through the same per-tap luminance shift as the per-pixel LUT schedule)."""
import numpy as np

from . import oracle


def project(K, D, H, cfg):
    """(px, py, valid) float64 [BH, BW]: raw-frame position sampled by every BEV pixel of one camera."""
    K, D, H = (np.asarray(a, np.float64) for a in (K, D, H))
    D = D.ravel()
    fw, fh, ss = cfg["FRAME_WIDTH"], cfg["FRAME_HEIGHT"], cfg["SIZE_SCALE"]
    bw, bh = cfg["BEV_WIDTH"], cfg["BEV_HEIGHT"]
    uw, uh = int(fw * ss), int(fh * ss)
    Kd = oracle.camera_mat_dst(K, fw, fh, cfg["FOCAL_SCALE"], ss)
    M = oracle.invert3x3(H)
    yy, xx = np.mgrid[0:bh, 0:bw].astype(np.float64)
    X = M[0, 0] * xx + M[0, 1] * yy + M[0, 2]
    Y = M[1, 0] * xx + M[1, 1] * yy + M[1, 2]
    Wd = M[2, 0] * xx + M[2, 1] * yy + M[2, 2]
    with np.errstate(divide="ignore", invalid="ignore"):
        u, v = X / Wd, Y / Wd
        valid = (Wd != 0) & (u >= 0) & (u <= uw - 1) & (v >= 0) & (v <= uh - 1)
        xn, yn = (u - Kd[0, 2]) / Kd[0, 0], (v - Kd[1, 2]) / Kd[1, 1]
        r = np.sqrt(xn * xn + yn * yn)
        theta = np.arctan(r)
        t2 = theta * theta
        t4 = t2 * t2
        t6 = t4 * t2
        t8 = t4 * t4
        theta_d = theta * (1 + D[0] * t2 + D[1] * t4 + D[2] * t6 + D[3] * t8)
        scale = np.where(r == 0, 1.0, theta_d / np.where(r == 0, 1.0, r))
        px = K[0, 0] * xn * scale + K[0, 2]
        py = K[1, 1] * yn * scale + K[1, 2]
    valid &= (px > -1.0) & (px < fw) & (py > -1.0) & (py < fh)
    return np.where(valid, px, 0.0), np.where(valid, py, 0.0), valid


def sample(img, px, py, valid):
    """fp64 bilinear, BORDER_CONSTANT 0 per tap, round half to even -> uint8 [BH, BW, 3]"""
    h, w = img.shape[:2]
    fx, fy = np.floor(px), np.floor(py)
    sx, sy = fx.astype(np.int64), fy.astype(np.int64)
    ax, ay = (px - fx)[..., None], (py - fy)[..., None]

    def tap(dx, dy):
        x, y = sx + dx, sy + dy
        ok = (x >= 0) & (x < w) & (y >= 0) & (y < h)
        t = img[np.clip(y, 0, h - 1), np.clip(x, 0, w - 1)].astype(np.float64)
        return np.where(ok[..., None], t, 0.0)
    top = (1.0 - ax) * tap(0, 0) + ax * tap(1, 0)
    bot = (1.0 - ax) * tap(0, 1) + ax * tap(1, 1)
    val = np.rint((1.0 - ay) * top + ay * bot)
    return np.where(valid[..., None], np.clip(val, 0, 255), 0).astype(np.uint8)


class AnalyticBevGenerator:
    """BevGenerator(blend)(front, back, left, right, car) with the analytic projection; masks and sums are the reference's."""

    def __init__(self, rig, cfg, blend=False):
        self.ref = oracle.RefBevGenerator(rig, cfg, blend=blend, balance=False)
        self.proj = [project(*rig[n], self.ref.cfg) for n in oracle.CAMERAS]

    def __call__(self, front, back, left, right, car=None):
        parts = [self.ref.apply_mask(i, sample(img, *self.proj[i])) for i, img in enumerate((front, back, left, right))]
        out = oracle.add_sat(parts[0], parts[1])
        out = oracle.add_sat(out, parts[2])
        out = oracle.add_sat(out, parts[3])
        if car is not None:
            out = oracle.add_sat(out, car)
        return out
