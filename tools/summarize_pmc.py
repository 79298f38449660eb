"""Per-launch HBM traffic of the stitch kernels from the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes collected by
tools/collect_profiles.sh: writes <dir>/rocprofv3_pmc_hbm_traffic.md and <dir>/hbm_traffic.json."""
import csv
import json
import os
import sys
from collections import defaultdict

STEPS, WARMUP = 3, 1
UNITS = {"direct_stitch_b256": 256, "blend_balance_b256": 256, "undistort_b64": 64}
# kernels whose reads are wide coalesced streams (16 B per lane): FETCH_SIZE tallies their 128-byte requests at 64 bytes on
# gfx950 (MI355X_MICROARCH.md, HBM section) -> doubled.  Calibration in this very run: k_vsum reads exactly
# 256 x 4 x 1280 x 960 x 3 B = 3775 MB per launch and FETCH_SIZE reports half of that.
WIDE_READERS = ("k_vsum", "k_gain", "k_reduce_psums")
PER_STEP = ("k_plan_all", "k_plan_staged", "k_plan_lean", "k_plan_empty", "k_stitch_", "k_vsum", "k_lum_", "k_reduce_psums", "k_gain", "k_remap")


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


def main(d):
    md = ["# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), `python bench.py --workload W --steps 3 --warmup 1 --no-cpu-baseline`", "",
          "KB per launch (= one bench step), per-step kernels only. FETCH_SIZE of the wide-stream readers (k_vsum, k_gain_lut, k_reduce_psums) is",
          "DOUBLED as MI355X_MICROARCH.md prescribes for gfx950 (128-byte requests tallied at 64 bytes; calibration in this run: k_vsum reads exactly",
          "3775 MB per launch). The stitch kernels request 64-byte sectors (4 lanes x 16 B), for which the guide gives no factor: they are reported",
          "as counted (x1; TCC_EA0_RDREQ x 64 B agreed in the round-1 check), i.e. a LOWER bound if the L2 merged neighbouring sectors into 128-byte",
          "requests. WRITE_SIZE is uncalibrated in the guide; the stitch kernels write 896 MB of pixels per launch and it reports 1.19x that.", ""]
    traffic = {"_comment": "HBM bytes per launch from separate rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes "
                           "(profiles/r01_final/rocprofv3_pmc_hbm_traffic.md, tools/collect_profiles.sh); bench.py copies the "
                           "figure of the workload it runs into roofline.traffic."}
    for w, units in UNITS.items():
        f, wr = os.path.join(d, "pmc_%s_FETCH_SIZE.csv" % w), os.path.join(d, "pmc_%s_WRITE_SIZE.csv" % w)
        if not (os.path.exists(f) and os.path.exists(wr)):
            continue
        fs, ws = kernel_sums(f), kernel_sums(wr)
        launches = STEPS + WARMUP
        md += ["## %s (%d units per launch)" % (w, units), "", "| kernel | FETCH_SIZE KB (wide readers x2) | WRITE_SIZE KB |", "|---|---|---|"]
        tf = tw = 0.0
        for k in sorted(set(fs) | set(ws)):
            if not any(p in k for p in PER_STEP):
                continue
            a, b = fs.get(k, 0.0) / launches, ws.get(k, 0.0) / launches
            if any(p in k for p in WIDE_READERS):
                a *= 2
            tf += a
            tw += b
            md.append("| `%s` | %.0f | %.0f |" % (k[:90], a, b))
        md += ["| **sum** | %.0f (%.0f MB) | %.0f (%.0f MB) |" % (tf, tf * 1024 / 1e6, tw, tw * 1024 / 1e6), ""]
        traffic[w] = {"fetch_bytes": int(tf * 1024), "write_bytes": int(tw * 1024), "units_per_launch": units, "round": 1}
    open(os.path.join(d, "rocprofv3_pmc_hbm_traffic.md"), "w").write("\n".join(md) + "\n")
    json.dump(traffic, open(os.path.join(d, "hbm_traffic.json"), "w"), indent=1)
    print("\n".join(md[-12:]))


if __name__ == "__main__":
    main(sys.argv[1])
