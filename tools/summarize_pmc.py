"""Per-launch HBM traffic of the stitch kernels from the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes collected by
tools/collect_profiles.sh: writes <dir>/rocprofv3_pmc_hbm_traffic.md and <dir>/hbm_traffic.json.

    python tools/summarize_pmc.py <dir>                  traffic tables (applies profiles/pmc_calibration.json when present)
    python tools/summarize_pmc.py --calibration <dir>    counter calibration on known byte counts (tools/calibrate_pmc.sh):
                                                         writes <dir>/calibration.md and <dir>/calibration.json
"""
import csv
import json
import os
import sys
from collections import defaultdict

STEPS, WARMUP = 3, 1
UNITS = {"direct_stitch_b256": 256, "blend_balance_b256": 256, "undistort_b64": 64, "blend_4k": 32, "blend_b256": 256}
# kernels whose reads are wide coalesced streams (16 B per lane): FETCH_SIZE tallies their 128-byte requests at 64 bytes on
# gfx950 (MI355X_MICROARCH.md, HBM section) -> doubled.  Calibration in this very run: k_vsum reads exactly
# 256 x 4 x 1280 x 960 x 3 B = 3775 MB per launch and FETCH_SIZE reports half of that.
WIDE_READERS = ("k_vsum", "k_gain", "k_reduce_psums")
# round 2: factors MEASURED on known byte counts in the kernels' own access shapes (tools/calibrate_pmc.sh ->
# profiles/pmc_calibration.json): FETCH_SIZE tallies every request at 64 bytes, so 128-byte line requests read 1/2 (x2.00),
# single 64-byte sectors x0.99, the pair-staged kernels' texel-group loads (16 B per lane, 12 B apart: mostly whole lines)
# x1.72; WRITE_SIZE is exact (x1.00 streams, x0.99 for the 8 x 96-byte tile stores).
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
try:
    CAL = json.load(open(os.path.join(ROOT, "profiles", "pmc_calibration.json")))["factors"]
except (OSError, ValueError, KeyError):
    CAL = None
GROUP_LOADERS = ("k_plan_units", "k_plan_unit_wide", "k_plan_all", "k_plan_pair", "k_plan_block", "k_plan_unit")   # the staged stitch kernels (texel-group loads)
PER_STEP = ("k_plan_units", "k_plan_all", "k_plan_unit", "k_plan_block", "k_plan_lean", "k_plan_empty", "k_stitch_", "k_vsum", "k_lum_", "k_reduce_psums", "k_gain", "k_remap")


def kernel_sums(path):
    """kernel name -> sum of the counter over all dispatches (KB for FETCH_SIZE / WRITE_SIZE)."""
    out = defaultdict(float)
    with open(path) as fh:
        for row in csv.DictReader(fh):
            name = row.get("Kernel_Name") or row.get("Kernel Name") or row.get("kernel_name")
            val = row.get("Counter_Value") or row.get("Counter Value") or row.get("counter_value")
            if name and val:
                out[name] += float(val)
    return out


def per_launch_traffic(fetch_csv, write_csv, launches):
    """(corrected FETCH KB, WRITE KB, rows) per launch of the per-step kernels in two counter files: the correction rules of this module
    (wide-stream readers x stream_read, the staged stitch kernels' group loads x group_loads and tile stores x tile_stores, k_lum_groups
    x gather_64B).  Shared by main() below and by bench.py's live traffic measurement."""
    fs, ws = kernel_sums(fetch_csv), kernel_sums(write_csv)
    tf = tw = 0.0
    rows = []
    for k in sorted(set(fs) | set(ws)):
        if not any(p in k for p in PER_STEP):
            continue
        a, b = fs.get(k, 0.0) / launches, ws.get(k, 0.0) / launches
        if any(p in k for p in WIDE_READERS):
            a *= CAL["stream_read"] if CAL else 2
        elif CAL and any(p in k for p in GROUP_LOADERS):
            a *= CAL["group_loads"]
            b *= CAL["tile_stores"]
        elif CAL and "k_lum_groups" in k:
            a *= CAL["gather_64B"]
        tf += a
        tw += b
        rows.append((k, a, b))
    return tf, tw, rows


def main(d):
    md = ["# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), `python bench.py --workload W --steps 3 --warmup 1 --no-cpu-baseline`", "",
          "KB per launch (= one bench step), per-step kernels only, CORRECTED with the factors measured on known byte counts in the kernels' own",
          "access shapes (profiles/r02/pmc_calibration.md, tools/calibrate_pmc.sh): FETCH_SIZE tallies every request at 64 bytes -> wide-stream readers",
          "(k_vsum, k_gain_lut, k_reduce_psums) x%.2f, the pair-staged stitch kernels' texel-group loads x%.2f, single-sector gathers x%.2f;" %
          ((CAL or {}).get("stream_read", 2.0), (CAL or {}).get("group_loads", 1.0), (CAL or {}).get("gather_64B", 1.0)),
          "WRITE_SIZE is exact (x%.2f for the 8 x 96-byte tile stores), so write traffic above the output bytes is real (partial-sector evictions)." %
          (CAL or {}).get("tile_stores", 1.0), ""]
    traffic = {"_comment": "HBM bytes per launch from separate rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes "
                           "(tools/collect_profiles.sh -> rocprofv3_pmc_hbm_traffic.md), corrected with the factors of "
                           "profiles/pmc_calibration.json (measured on known byte counts); bench.py measures the same figure live "
                           "(roofline.traffic) and falls back to this file when rocprofv3 is not available."}
    for w, units in UNITS.items():
        f, wr = os.path.join(d, "pmc_%s_FETCH_SIZE.csv" % w), os.path.join(d, "pmc_%s_WRITE_SIZE.csv" % w)
        if not (os.path.exists(f) and os.path.exists(wr)):
            continue
        launches = STEPS + WARMUP
        md += ["## %s (%d units per launch)" % (w, units), "", "| kernel | FETCH_SIZE KB (wide readers x2) | WRITE_SIZE KB |", "|---|---|---|"]
        tf, tw, rows = per_launch_traffic(f, wr, launches)
        for k, a, b in rows:
            md.append("| `%s` | %.0f | %.0f |" % (k[:90], a, b))
        md += ["| **sum** | %.0f (%.0f MB) | %.0f (%.0f MB) |" % (tf, tf * 1024 / 1e6, tw, tw * 1024 / 1e6), ""]
        traffic[w] = {"fetch_bytes": int(tf * 1024), "write_bytes": int(tw * 1024), "units_per_launch": units, "round": ROUND,
                      "corrected": bool(CAL)}
    open(os.path.join(d, "rocprofv3_pmc_hbm_traffic.md"), "w").write("\n".join(md) + "\n")
    json.dump(traffic, open(os.path.join(d, "hbm_traffic.json"), "w"), indent=1)
    print("\n".join(md[-12:]))


ROUND = int(os.environ.get("BEVW_ROUND", "6"))   # the round the collection belongs to (a label in hbm_traffic.json)


def dispatches(path):
    """[(kernel name, counter value)] in dispatch order, runtime helper kernels (fills / copies) dropped."""
    rows = []
    with open(path) as fh:
        for row in csv.DictReader(fh):
            name = row.get("Kernel_Name") or ""
            if name.startswith("__amd_rocclr") or not name:
                continue
            rows.append((int(row.get("Dispatch_Id") or len(rows)), name, float(row.get("Counter_Value") or 0.0)))
    rows.sort()
    return [(n, v) for _, n, v in rows]


def calibration(d):
    """tools/hbm_stream --calib launches each kernel once and prints `CALIB <label> read_bytes R write_bytes W`; the two
    counter passes list the same dispatches in the same order."""
    known = []
    for line in open(os.path.join(d, "known_bytes.txt")):
        if line.startswith("CALIB"):
            label = line[6:].split("read_bytes")[0].strip()
            rd = float(line.split("read_bytes")[1].split()[0])
            wr = float(line.split("write_bytes")[1].split()[0])
            known.append((label, rd, wr))
    fetch, write = dispatches(os.path.join(d, "FETCH_SIZE.csv")), dispatches(os.path.join(d, "WRITE_SIZE.csv"))
    if not (len(fetch) == len(write) == len(known)):
        raise SystemExit("dispatch count mismatch: %d known, %d FETCH_SIZE rows, %d WRITE_SIZE rows" % (len(known), len(fetch), len(write)))
    md = ["# FETCH_SIZE / WRITE_SIZE calibrated on known byte counts (`tools/calibrate_pmc.sh`, one launch per kernel, 2 GB buffers)", "",
          "counter bytes = value x 1024 (rocprofv3 reports KB). factor = known bytes / counter bytes = what a reading of that access shape",
          "has to be multiplied by. The guide's gfx950 note (FETCH_SIZE reads 1/2 of a wide coalesced stream) is the `stream read` row.", "",
          "| access shape | known read MB | FETCH_SIZE MB | read factor | known write MB | WRITE_SIZE MB | write factor |", "|---|---|---|---|---|---|---|"]
    out = {}
    for (label, rd, wr), (kn, fv), (_, wv) in zip(known, fetch, write):
        fb, wb = fv * 1024.0, wv * 1024.0
        rf = rd / fb if rd > 0 and fb > 0 else None
        wf = wr / wb if wr > 0 and wb > 0 else None
        md.append("| %s (`%s`) | %.0f | %.0f | %s | %.0f | %.0f | %s |" % (label, kn.split("(")[0][-40:], rd / 1e6, fb / 1e6,
                  "%.3f" % rf if rf else "-", wr / 1e6, wb / 1e6, "%.3f" % wf if wf else "-"))
        out[label] = {"kernel": kn.split("(")[0], "read_factor": rf, "write_factor": wf}
    pick = lambda key, field: next((v[field] for k, v in out.items() if key in k and v[field]), None)
    factors = {"stream_read": pick("stream read", "read_factor"), "stream_write": pick("stream write", "write_factor"),
               "group_loads": pick("group loads", "read_factor"), "tile_stores": pick("tile stores", "write_factor"),
               "gather_64B": pick("gather 9 x 64 B random : 0", "read_factor"), "gather_128B": pick("gather 9 x 128 B random : 0", "read_factor")}
    md += ["", "Factors applied by `tools/summarize_pmc.py` to the stitch kernels (group loads / tile stores) and to the wide-stream kernels:", "",
           "```", json.dumps(factors, indent=1), "```", ""]
    open(os.path.join(d, "calibration.md"), "w").write("\n".join(md) + "\n")
    json.dump({"factors": factors, "shapes": out}, open(os.path.join(d, "calibration.json"), "w"), indent=1)
    print("\n".join(md))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--calibration":
        calibration(sys.argv[2])
    else:
        main(sys.argv[1])
