"""profiles/jpeg_valu.json from rocprofv3 --pmc SQ_INSTS_VALU passes of the JPEG workloads (tools/collect_profiles.sh):

    python tools/jpeg_valu.py <dir with pmc_valu_<workload>.csv files>  ->  <dir>/jpeg_valu.json (+ a per-kernel table on stdout)

Wave-level VALU instructions per UNIT of every workload: the sum over all dispatches of the step's kernels divided by the number of
steps the process ran (counted through a kernel that runs once per step) and by the units of a step.  bench.py turns the figure into
roofline.achieved of the JPEG lines (bound "valu_issue")."""
import csv
import json
import os
import sys
from collections import defaultdict

# workload -> (kernel-name prefixes of one step, a kernel with a known number of dispatches per step, that number, units per step)
# (a decode batch of 256 files runs as 2 slices, each with its own k_jpeg_subs: bevwarp_jpeg.hip, bevw_jpeg_decode_run_device)
SPEC = {
    "jpeg_decode_b64": (("k_jpeg_",), "k_jpeg_subs", 2, 256),
    "jpeg_decode_b64_repo": (("k_jpeg_",), "k_jpeg_subs", 2, 256),
    "jpeg_encode_b64": (("k_jenc_",), "k_jenc_scan", 1, 64),
    "jpeg_bev_jpeg_b64": (("k_jpeg_", "k_jenc_", "k_plan_", "k_stitch_plan"), "k_jenc_scan", 1, 64),
}


def short(name):
    n = name.split("(")[0]
    for p in ("void ", "bevw::jpg::", "bevw::"):
        n = n.replace(p, "")
    return n.split("<")[0].strip()


def main(d):
    out = {"_comment": "wave-level VALU instructions per unit (SQ_INSTS_VALU summed over the kernels of one step / units per step) from rocprofv3 --pmc "
                       "passes of `bench.py --workload W --steps 3 --warmup 1 --no-cpu-baseline` (tools/collect_profiles.sh); static figures of the round they "
                       "were collected in; bench.py: roofline.bound = valu_issue"}
    for w, (prefixes, marker, per_step_calls, units) in SPEC.items():
        path = os.path.join(d, "pmc_valu_%s.csv" % w)
        if not os.path.exists(path):
            continue
        sums, calls = defaultdict(float), defaultdict(int)
        with open(path) as fh:
            for row in csv.DictReader(fh):
                if (row.get("Counter_Name") or row.get("Counter Name")) != "SQ_INSTS_VALU":
                    continue
                k = short(row.get("Kernel_Name") or row.get("Kernel Name"))
                sums[k] += float(row.get("Counter_Value") or row.get("Counter Value"))
                calls[k] += 1
        steps = calls.get(marker, 0) // per_step_calls
        if not steps:
            continue
        per_step = {k: v / steps for k, v in sums.items() if k.startswith(prefixes)}
        total = sum(per_step.values())
        out[w] = {"valu_wave_insts_per_unit": total / units, "units_per_step": units, "steps_profiled": steps, "round": 4,
                  "source": "profiles/r05_final/pmc_valu_%s.csv (rocprofv3 --pmc SQ_INSTS_VALU)" % w,
                  "per_kernel_per_step": {k: round(v) for k, v in sorted(per_step.items(), key=lambda kv: -kv[1])}}
        print("%s: %.0f M wave-level VALU instructions per step (%d steps profiled), %.0f per unit" % (w, total / 1e6, steps, total / units))
        for k, v in sorted(per_step.items(), key=lambda kv: -kv[1]):
            print("    %-28s %8.1f M  (%d dispatches per step)" % (k, v / 1e6, calls[k] // steps))
    json.dump(out, open(os.path.join(d, "jpeg_valu.json"), "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1])
