"""Workload definitions shared by bench.py and the tests: the repo rig's calibration constants, the synthetic
BASELINE.json configurations derived from it (SURVEY.md 8d / BASELINE.md section 3) and the seeded frame generator.
Host-side bookkeeping only -- nothing here touches pixels of the product path.
"""
from __future__ import annotations

import numpy as np

CAMERA_NAMES = ("front", "back", "left", "right")
SEED = 3045  # 0x0BE5

# K (3x3), D (4), H (3x3), float64, exactly the values of the reference's
# SurroundBirdEyeView/data/<cam>/camera_<cam>_{K,D,H}.npy (loaded at surroundBEV.py:83-85); tests/test_workloads.py checks
# them against tests/golden/repo_rig.npz.
_REPO_RIG = {
    'front': (
        [350.4931893001142, 0.0, 647.6297467576265, 0.0, 352.43072872484805, 513.5196785119657, 0.0, 0.0, 1.0],
        [-0.03367245449576437, 0.015380779195912842, -0.018654590946883556, 0.0058128945633924185],
        [-0.04794763202074084, -0.5818155471242885, 590.0520678923119, 0.013279591563462981, -0.39037264999919646, 362.27094297250517, 3.988296657965681e-05, -0.0011579141685543123, 1.0],
    ),
    'back': (
        [349.38488390073064, 0.0, 604.8877591311758, 0.0, 347.74107181362274, 530.5836779187023, 0.0, 0.0, 1.0],
        [-0.03127288805593267, 0.00011957979989533713, -0.0011784280539928825, -0.00019601823489868008],
        [0.06388897981449383, -0.5707646115263133, 415.3248311292778, -0.005319575032398037, -0.7476912588025478, 631.84343812624, -8.781973293017248e-06, -0.0011359946058636821, 1.0],
    ),
    'left': (
        [346.24479038664373, 0.0, 634.0605869340807, 0.0, 345.72196256377504, 507.1511336952081, 0.0, 0.0, 1.0],
        [-0.020771033532929598, -0.012486337517085708, 0.005252880419078808, -0.0012957901124682762],
        [-0.0007332862687965804, -0.48755573296372107, 488.35003847942846, 0.10637143971079516, -0.49503688503434246, 317.59089723215675, -4.025887072755806e-06, -0.0010837365820689183, 1.0],
    ),
    'right': (
        [346.5378151679977, 0.0, 633.0762288849737, 0.0, 345.15910333237264, 517.0740406903644, 0.0, 0.0, 1.0],
        [-0.028161366852994228, -0.0011162652180589155, -0.0016547436039781613, 6.0719100100213355e-05],
        [0.026679181239480345, -0.637598682625524, 531.4247579128246, -0.08614891447231514, -0.5239027523583694, 609.5358592656677, 5.8195140734011604e-05, -0.0011318945342837438, 0.9999999999999999],
    ),
}


def repo_rig():
    """name -> (K 3x3, D 4x1, H 3x3) of the reference's sample rig (config R: 1280x1024 -> 1000x1000)."""
    return {n: (np.array(K, np.float64).reshape(3, 3), np.array(D, np.float64).reshape(4, 1),
                np.array(H, np.float64).reshape(3, 3)) for n, (K, D, H) in _REPO_RIG.items()}


# args-style dictionaries (the reference's argparse Namespace fields, surroundBEV.py:6-17)
CONFIG_R = dict(FRAME_WIDTH=1280, FRAME_HEIGHT=1024, BEV_WIDTH=1000, BEV_HEIGHT=1000, CAR_WIDTH=250, CAR_HEIGHT=400,
                FOCAL_SCALE=1.0, SIZE_SCALE=2.0)
CONFIG_S = dict(FRAME_WIDTH=1280, FRAME_HEIGHT=960, BEV_WIDTH=1080, BEV_HEIGHT=1080, CAR_WIDTH=270, CAR_HEIGHT=432,
                FOCAL_SCALE=1.0, SIZE_SCALE=2.0)
CONFIG_4K = dict(FRAME_WIDTH=3840, FRAME_HEIGHT=2160, BEV_WIDTH=1080, BEV_HEIGHT=1080, CAR_WIDTH=270, CAR_HEIGHT=432,
                 FOCAL_SCALE=1.0, SIZE_SCALE=2.0)
CONFIG_UNDISTORT = dict(FRAME_WIDTH=1280, FRAME_HEIGHT=960, FOCAL_SCALE=0.5, SIZE_SCALE=1.0)


def rig_s():
    """BASELINE synthetic rig (1280x960 -> 1080x1080): the repo frames centre-cropped 1024 -> 960 rows.
    K_S = K with cy -= 32; H_S = diag(1.08, 1.08, 1) . H . T(0, +64) (T maps new undistorted y to the old one)."""
    S = np.diag([1.08, 1.08, 1.0])
    T = np.array([[1.0, 0.0, 0.0], [0.0, 1.0, 64.0], [0.0, 0.0, 1.0]])
    out = {}
    for n, (K, D, H) in repo_rig().items():
        Ks = K.copy()
        Ks[1, 2] -= 32.0
        out[n] = (Ks, D.copy(), S @ H @ T)
    return out


def rig_4k():
    """4K rig (3840x2160 -> 1080x1080): frames scaled x3 then centre-cropped 3072 -> 2160 rows.
    fx, fy, cx x3; cy = 3 cy - 456; H_4K = diag(1.08,1.08,1) . H . [[1/3,0,0],[0,1/3,304],[0,0,1]]."""
    S = np.diag([1.08, 1.08, 1.0])
    A = np.array([[1.0 / 3, 0.0, 0.0], [0.0, 1.0 / 3, 304.0], [0.0, 0.0, 1.0]])
    out = {}
    for n, (K, D, H) in repo_rig().items():
        Kk = K.copy()
        Kk[0, 0] *= 3; Kk[1, 1] *= 3; Kk[0, 2] *= 3
        Kk[1, 2] = 3 * K[1, 2] - 456.0
        out[n] = (Kk, D.copy(), S @ H @ A)
    return out


def undistort_calibration():
    """Config 2: K/D of `front` with the 960-row crop (cy -= 32)."""
    K, D, _ = rig_s()["front"]
    return K, D


def synthetic_frames(n_sets: int, width: int, height: int, seed: int = SEED, kind: str = "smooth") -> np.ndarray:
    """uint8 [n_sets, 4, height, width, 3] BGR.
    smooth:   sum of 6 random 2-D sinusoids per channel (amplitude 90 in total, offset 128) + uniform noise +-16
    random:   every byte uniform in [0, 255]  (worst case for parity)
    constant: one random colour per frame       (known answer away from borders)"""
    rng = np.random.default_rng(seed)
    if kind == "random":
        return rng.integers(0, 256, (n_sets, 4, height, width, 3), dtype=np.uint8)
    if kind == "constant":
        col = rng.integers(0, 256, (n_sets, 4, 1, 1, 3), dtype=np.uint8)
        return np.ascontiguousarray(np.broadcast_to(col, (n_sets, 4, height, width, 3)))
    if kind != "smooth":
        raise ValueError(kind)
    out = np.empty((n_sets, 4, height, width, 3), np.uint8)
    yy = np.arange(height, dtype=np.float64) / height
    xx = np.arange(width, dtype=np.float64) / width
    for s in range(n_sets):
        for c in range(4):
            field = np.empty((height, width, 3), np.float32)
            for ch in range(3):
                fx, fy = rng.uniform(0.5, 6.0, (2, 6))
                ph = rng.uniform(0, 2 * np.pi, 6)
                ax = 2 * np.pi * fx[:, None] * xx[None, :] + ph[:, None]   # [6, W]
                by = 2 * np.pi * fy[:, None] * yy[None, :]                 # [6, H]
                # sum_k sin(ax_k + by_k) = [cos by; sin by]^T [sin ax; cos ax]   (one small GEMM per channel)
                A = np.concatenate([np.cos(by), np.sin(by)]).astype(np.float32)
                B = np.concatenate([np.sin(ax), np.cos(ax)]).astype(np.float32)
                field[:, :, ch] = 128.0 + 15.0 * np.einsum('kh,kw->hw', A, B)  # (BLAS threads thrash in small containers)
            field += rng.integers(0, 33, (height, width, 3), dtype=np.uint8).astype(np.float32) - 16.0
            out[s, c] = np.clip(np.rint(field), 0, 255).astype(np.uint8)
    return out


# Algorithmic (compulsory) bytes per unit of work: SURVEY.md 8(d) / BASELINE.md section 3 -- unique sampled source
# texels x 3 B + output bytes; static tables excluded (shared by every frame of a batch).
ALGORITHMIC_BYTES = {
    "repo_direct": 4_949_148,          # config 1: touched 1,949,148 + out 3,000,000
    "undistort_b64": 5_421_912,        # config 2: touched 1,735,512 + out 3,686,400
    "direct_stitch_b256": 5_532_357,   # config 3: touched 2,033,157 + out 3,499,200
    "blend_b256": 5_669_538,           # S blend only: touched 2,170,338 + out 3,499,200
    "blend_balance_b256": 22_585_476,  # config 4: V-mean pass 14,745,600 + 2 x touched 2,170,338 + out 3,499,200
    "blend_4k": 11_810_991,            # config 5 (blend only): touched 8,311,791 + out 3,499,200
    "blend_4k_camera_shard": 11_810_991,  # same frames, camera per GPU (exchange bytes are overhead, not counted)
    "direct_stitch_analytic_f32_b64": 5_532_357,   # config 3's bytes: the analytic modes touch (nearly) the same texels
    "direct_stitch_analytic_f64_b64": 5_532_357,
    "direct_stitch_analytic_perpixel_b64": 5_532_357,
}

# The same workloads at 64-byte SECTOR granularity: the unique 64-byte segments of the frames the sampled texels touch (SURVEY.md B.2,
# "sector-granular B") + the output bytes -- the least a memory system that moves whole sectors can move, whatever the kernel.  A strongly
# minifying map (the 4K rig: 3.5 x) uses 44 % of every sector it has to fetch, so its roofline fraction on ALGORITHMIC bytes (0.23) reads
# worse than the kernel is; bench.py reports both (roofline.frac_sector_floor) so that the figure is interpretable.
SECTOR_GRANULAR_BYTES = {
    "direct_stitch_b256": 2_635_392 + 3_499_200,
    "blend_b256": 2_774_400 + 3_499_200,
    "blend_4k": 18_937_216 + 3_499_200,
    "blend_4k_camera_shard": 18_937_216 + 3_499_200,
    "direct_stitch_analytic_f32_b64": 2_635_392 + 3_499_200,
    "direct_stitch_analytic_f64_b64": 2_635_392 + 3_499_200,
    "direct_stitch_analytic_perpixel_b64": 2_635_392 + 3_499_200,
}
