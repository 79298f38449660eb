"""L1 <-> L2 request arithmetic of the stitch for a given tiling, from the LUTs alone (CPU, no GPU needed).

The stitch kernels are bound by the NUMBER of vector-L1 -> L2 requests per step (profiles/r02/sweeps.log: the step time
follows reads + writes at ~80 G requests/s whatever their size), so the schedule question is: how many 128-byte source
lines and 64-byte destination sectors does one staging unit touch?  This script counts them for

  * per-wave tiles (what bevw_pair.h does): every wave fetches the lines of its own 256-pixel tile,
  * block tiles: the waves of a block share ONE staged footprint (lines counted once per block tile),

on the direct-stitch masks of a bench configuration (default: config S = BASELINE config 3).  Output: requests per
256 output pixels, to be compared with the measured 21.5 read + 20.8 write requests per tile-frame of the 32 x 8 schedule.

    python tools/analyze_requests.py [--config S|R|4K] [--blend]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cameracalibration_amd import workloads as W  # noqa: E402
from oracle import oracle  # noqa: E402  (analysis tool: the oracle only supplies the LUTs)


def tables(cfg, rig, blend):
    gen = oracle.RefBevGenerator(rig, cfg, blend=blend, balance=False)
    fw, fh = cfg["FRAME_WIDTH"], cfg["FRAME_HEIGHT"]
    out = []
    for ci, cam in enumerate(gen.cameras):
        m1 = cam.bev_maps[0]
        mask = gen.masks[ci]
        if mask.ndim == 3:
            mask = mask[..., 0]
        sx, sy = m1[..., 0].astype(np.int64), m1[..., 1].astype(np.int64)
        ok = (mask != 0) & (sx >= 0) & (sx < fw - 1) & (sy >= 0) & (sy < fh - 1)
        out.append((sx, sy, ok))
    return out, fw, fh


def count(tabs, fw, fh, bw, bh, tw, th):
    """distinct 128-byte lines, 64-byte sectors and 16-byte texel groups over the tiles of tw x th pixels"""
    row_bytes = fw * 3
    nt_x, nt_y = (bw + tw - 1) // tw, (bh + th - 1) // th
    yy, xx = np.mgrid[0:bh, 0:bw]
    tile = (yy // th) * nt_x + xx // tw
    lines = sectors = groups = 0
    hist = np.zeros(nt_x * nt_y, np.int64)
    for ci, (sx, sy, ok) in enumerate(tabs):
        t = tile[ok]
        x, y = sx[ok], sy[ok]
        base = ci * fh * row_bytes
        # a footprint = texels x, x+1 on rows y, y+1: bytes [3x, 3x+6) of both rows
        keys_l, keys_s, keys_g = [], [], []
        for dy in (0, 1):
            for b in (0, 5):
                a = base + (y + dy) * row_bytes + 3 * x + b
                keys_l.append(t * (1 << 34) + a // 128)
                keys_s.append(t * (1 << 34) + a // 64)
            # texel groups of 4 (12 bytes): pairs (x, x+1) live in group x // 4 (the group carries texel 4g+4 as well)
            keys_g.append(t * (1 << 34) + (base + (y + dy) * row_bytes) // 12 + x // 4)
        ul = np.unique(np.concatenate(keys_l))
        lines += ul.size
        sectors += np.unique(np.concatenate(keys_s)).size
        ug = np.unique(np.concatenate(keys_g))
        groups += ug.size
        np.add.at(hist, (ug >> 34).astype(np.int64), 1)
    return lines, sectors, groups, hist


def write_sectors(bw, bh, tw, th):
    """64-byte sectors touched by the row segments of tw-pixel-wide tiles (rows of bw * 3 bytes), total and partial"""
    total = partial = 0
    for y in range(bh):
        for x0 in range(0, bw, tw):
            a0, a1 = (y * bw + x0) * 3, (y * bw + min(bw, x0 + tw)) * 3
            s0, s1 = a0 // 64, (a1 - 1) // 64
            total += s1 - s0 + 1
            partial += (a0 % 64 != 0) + (a1 % 64 != 0 and (s1 != s0 or a0 % 64 == 0))
    return total, partial


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="S", choices=["S", "R", "4K"])
    ap.add_argument("--blend", action="store_true")
    a = ap.parse_args()
    cfg, rig = {"S": (W.CONFIG_S, W.rig_s()), "R": (W.CONFIG_R, W.repo_rig()), "4K": (W.CONFIG_4K, W.rig_4k())}[a.config]
    tabs, fw, fh = tables(cfg, rig, a.blend)
    bw, bh = cfg["BEV_WIDTH"], cfg["BEV_HEIGHT"]
    npx = sum(int(ok.sum()) for _, _, ok in tabs)
    print("config %s%s: %d x %d BEV, %d contributing pixels (with multiplicity), frames %d x %d" % (a.config, " blend" if a.blend else "", bw, bh, npx, fw, fh))
    print("%-22s %10s %10s %10s %12s   per 256 px: lines sectors groups | write sectors (partial)" % ("staging unit", "lines", "sectors", "groups", "max groups"))
    for tw, th in [(32, 8), (64, 4), (16, 16), (32, 32), (64, 16), (64, 32), (128, 16), (128, 32), (64, 64), (128, 64), (1080, 8)]:
        l, s, g, hist = count(tabs, fw, fh, bw, bh, tw, th)
        k = 256.0 / (bw * bh)
        print("%-22s %10d %10d %10d %12d   %6.1f %6.1f %6.1f" % ("%d x %d" % (tw, th), l, s, g, hist.max(), l * k, s * k, g * k))
    for tw in (32, 64, 128, 256):
        t, p = write_sectors(bw, bh, tw, 1)
        print("row segments of %3d px: %.1f write sectors per 256 px, %.1f of them partial" % (tw, t * 256.0 / (bw * bh), p * 256.0 / (bw * bh)))


if __name__ == "__main__":
    main()
