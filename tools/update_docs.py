"""Rewrite the measured tables of DESIGN.md / BASELINE.md / README.md from profiles/r01_final/bench_*.json and
profiles/hbm_traffic.json (run after tools/collect_profiles.sh and copying its output into profiles/)."""
import csv
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles", "r01_final")
W = ["direct_stitch_b256", "blend_b256", "blend_balance_b256", "undistort_b64", "blend_4k"]
d = {w: json.load(open(os.path.join(P, "bench_%s.json" % w))) for w in W}
t = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
ALG = {"direct_stitch_b256": 5532357 * 256, "blend_balance_b256": 22585476 * 256, "undistort_b64": 5421912 * 64}


def fmt(v):
    return format(round(v), ",")


def cpu(w):
    return d[w]["cpu_baseline"]["value"]


def ratio(w):
    return (t[w]["fetch_bytes"] + t[w]["write_bytes"]) / ALG[w]


def kernel_avg_us():
    with open(os.path.join(P, "rocprofv3_kernel_stats_direct_stitch_b256.csv")) as fh:
        for r in csv.DictReader(fh):
            if "k_plan_all" in r["Name"]:
                return float(r["AverageNs"]) / 1e3
    return float("nan")


def replace_between(s, start, end, new):
    i0, i1 = s.index(start), s.index(end)
    return s[:i0] + new + s[i1:]


def design():
    p = os.path.join(ROOT, "DESIGN.md")
    s = open(p).read()

    def row(name, w, unit):
        x, main = d[w], w == "direct_stitch_b256"
        v, fr = fmt(x["value"]) + " " + unit, "%.3f" % x["roofline"]["frac"]
        return "| %s | %s | %.3f | %s | %.1f %s (%d) | %dx |" % (name, "**" + v + "**" if main else v, x["roofline"]["kernel_ms"],
                                                                "**" + fr + "**" if main else fr, cpu(w), unit,
                                                                x["cpu_baseline"]["cores"], round(x["value"] / cpu(w)))
    rows = [row("config 3 direct stitch, batch 256", "direct_stitch_b256", "frames/s"), row("blend only, batch 256", "blend_b256", "frames/s"),
            row("config 4 blend + balance, batch 256", "blend_balance_b256", "frames/s"), row("config 2 undistort, batch 64", "undistort_b64", "images/s"),
            row("config 5 geometry (4K blend, 1 GPU, batch 32)", "blend_4k", "frames/s")]
    ds, bb, ud = t["direct_stitch_b256"], t["blend_balance_b256"], t["undistort_b64"]
    text = "\n".join(rows) + "\n\n" + (
        "The numbers move by up to +-5 %% from box to box (config 3 between 353 k and 402 k frames/s over this round's runs of the same\n"
        "kernels); the table is one `tools/collect_profiles.sh` run on one box.\n\n"
        "rocprofv3 agrees with the HIP-event numbers: the step is ONE kernel, `k_plan_all<8,false,false>`, whose average in\n"
        "`rocprofv3_kernel_stats_direct_stitch_b256.csv` is %.0f us against `roofline.kernel_ms` = %.3f ms of the un-profiled run\n"
        "(per class, from the per-class launches of an earlier build: staged singles 299 us, gather singles 283 us, seam classes\n"
        "37 + 27 us, empty tiles 21 us). HBM traffic per launch (`rocprofv3_pmc_hbm_traffic.md`, separate FETCH_SIZE / WRITE_SIZE\n"
        "passes): config 3 %.0f + %.0f MB = %.2f x the 1.416 GB algorithmic bytes; config 2 %.0f + %.0f MB = %.2f x; config 4\n"
        "%.0f + %.0f MB = %.2f x the gather-twice accounting (the built store-and-rescale schedule writes the pre-gain BEV and the\n"
        "luminance-shifted texel groups once more) -- the kernels do not waste bandwidth, they under-use it.\n\n") % (
        kernel_avg_us(), d["direct_stitch_b256"]["roofline"]["kernel_ms"], ds["fetch_bytes"] / 1e6, ds["write_bytes"] / 1e6, ratio("direct_stitch_b256"),
        ud["fetch_bytes"] / 1e6, ud["write_bytes"] / 1e6, ratio("undistort_b64"), bb["fetch_bytes"] / 1e6, bb["write_bytes"] / 1e6, ratio("blend_balance_b256"))
    s = replace_between(s, "| config 3 direct stitch, batch 256 | **", "Round-1 progression of config 3", text)
    open(p, "w").write(s)


def baseline():
    p = os.path.join(ROOT, "BASELINE.md")
    s = open(p).read()
    x = d
    new = ("| CPU oracle (reference op order, OpenMP), config 3 direct stitch | %.1f frames/s | 16 threads of the GPU box's host | `bench.py` `cpu_baseline`, kind \"port\" (cv2 itself is not installable) |\n"
           "| CPU oracle, blend only / config 4 blend+balance / config 2 undistort | %.1f / %.1f frames/s / %s images/s | 16 threads | same |\n"
           "| MI355X, config 3 direct stitch, batch 256 | **%s frames/s** (%.3f ms per launch) | 1 GPU | roofline frac %.3f of 8 TB/s on compulsory bytes; measured HBM traffic %.2fx compulsory; 353 k - 402 k across boxes |\n"
           "| MI355X, blend only, batch 256 | %s frames/s | 1 GPU | frac %.3f |\n"
           "| MI355X, config 4 blend + balance, batch 256 | %s frames/s | 1 GPU | frac %.3f |\n"
           "| MI355X, config 2 undistort, batch 64 | %s images/s | 1 GPU | frac %.3f |\n"
           "| MI355X, config 5 geometry (4K blend) on one GPU, batch 32 | %s frames/s | 1 GPU | frac %.3f; the camera-per-GPU form (`bench.py --workload blend_4k_camera_shard`) is built and bit-exact, its RCCL transport unmeasured (1-GPU boxes) |\n") % (
        cpu("direct_stitch_b256"), cpu("blend_b256"), cpu("blend_balance_b256"), fmt(cpu("undistort_b64")),
        fmt(x["direct_stitch_b256"]["value"]), x["direct_stitch_b256"]["roofline"]["kernel_ms"], x["direct_stitch_b256"]["roofline"]["frac"], ratio("direct_stitch_b256"),
        fmt(x["blend_b256"]["value"]), x["blend_b256"]["roofline"]["frac"], fmt(x["blend_balance_b256"]["value"]), x["blend_balance_b256"]["roofline"]["frac"],
        fmt(x["undistort_b64"]["value"]), x["undistort_b64"]["roofline"]["frac"], fmt(x["blend_4k"]["value"]), x["blend_4k"]["roofline"]["frac"])
    s = replace_between(s, "| CPU oracle (reference op order, OpenMP), config 3 direct stitch |", "| Achievable HBM ceiling", new)
    open(p, "w").write(s)


def readme():
    p = os.path.join(ROOT, "README.md")
    s = open(p).read()
    x = d
    new = ("* 1 MI355X, batch 256, 4 x 1280x960 -> 1080x1080: **%.1f k stitched frames/s** direct (%.0f %% of the 8 TB/s roofline on\n"
           "  compulsory bytes), %d k blend, %d k blend+balance, %d k undistort images/s, %d k frames/s on the 4K rig; CPU oracle on 16\n"
           "  host threads: %.0f frames/s.  Details, profiles and the bound analysis: `DESIGN.md`, `profiles/`.\n\n") % (
        x["direct_stitch_b256"]["value"] / 1e3, x["direct_stitch_b256"]["roofline"]["frac"] * 100, round(x["blend_b256"]["value"] / 1e3),
        round(x["blend_balance_b256"]["value"] / 1e3), round(x["undistort_b64"]["value"] / 1e3), round(x["blend_4k"]["value"] / 1e3), cpu("direct_stitch_b256"))
    s = replace_between(s, "* 1 MI355X, batch 256", "Build: `python __graft_entry__.py`", new)
    open(p, "w").write(s)


if __name__ == "__main__":
    design()
    baseline()
    readme()
    print("docs updated from", P)
