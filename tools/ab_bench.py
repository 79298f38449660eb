"""Round-robin A/B of bench.py variants on ONE box (boxes differ by +-7 %, and one box drifts by several % between
back-to-back runs: only interleaved repeats of every variant inside one gpurun call compare).

    python tools/ab_bench.py --workload direct_stitch_b256 --reps 5 label1:ENV1=a,ENV2=b label2:BEVW_LIB_PATH=build_var/x.so ...

Prints per variant the min / median / max over the repeats of the per-step HIP-event median (roofline.kernel_ms_median).
"""
import argparse
import json
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="direct_stitch_b256")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--bench-args", default="", help="extra bench.py arguments for every variant, e.g. '--placements 2 --single-layout'")
    ap.add_argument("variants", nargs="+")
    a = ap.parse_args()
    variants = []
    for v in a.variants:
        label, _, envs = v.partition(":")
        env = dict(e.split("=", 1) for e in envs.split(",") if e)
        variants.append((label, env))
    res = {label: [] for label, _ in variants}
    for rep in range(a.reps):
        for label, env in variants:
            e = dict(os.environ)
            e.update(env)
            out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", a.workload, "--steps", str(a.steps), "--warmup", "5",
                                  "--no-cpu-baseline", "--no-f4", "--no-live-traffic"] + a.bench_args.split(), env=e, capture_output=True, text=True, timeout=300).stdout.strip().splitlines()
            try:
                d = json.loads(out[-1])
                res[label].append(d["roofline"].get("kernel_ms_median") or d["roofline"]["kernel_ms"])
            except (IndexError, ValueError, KeyError):
                res[label].append(float("nan"))
    for label, _ in variants:
        v = [x for x in res[label] if x == x]
        if v:
            print("AB %-18s %-28s min %.4f median %.4f max %.4f ms  (%d runs)" % (a.workload, label, min(v), statistics.median(v), max(v), len(v)))
        else:
            print("AB %-18s %-28s FAILED" % (a.workload, label))


if __name__ == "__main__":
    main()
