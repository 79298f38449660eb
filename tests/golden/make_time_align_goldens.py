#!/usr/bin/env python3
"""Golden vectors for cameracalibration_amd/Tools/timeAlign.py from the REFERENCE's own Tools/timeAlign.py.

Run in the build container only (imports /root/reference, which does not exist on the GPU box):

    python tests/golden/make_time_align_goldens.py

Writes tests/golden/time_align.json: seeded random timestamp scenarios (jitter, dropped and extra frames, late starters)
with the groups / camera order the reference's align_time returns (Tools/timeAlign.py:18-72).
"""
import importlib.util
import json
import os
import random
import sys

REF = os.environ.get("BEVW_REFERENCE_ROOT", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "time_align.json")


def main() -> int:
    spec = importlib.util.spec_from_file_location("ref_time_align", os.path.join(REF, "Tools", "timeAlign.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    rng = random.Random(20260924)
    cases = []
    for n in range(60):
        frames = rng.randint(3, 40)
        period = rng.choice([0.05, 0.1, 0.25, 0.5])
        thresh = rng.choice([0.02, 0.05, 0.1])
        start = rng.uniform(0.0, 1000.0)
        cams = ["front", "back", "left", "right"][: rng.randint(2, 4)]
        rng.shuffle(cams)
        td = {}
        for c in cams:
            stamps = []
            skip = rng.randint(0, 3)          # late starter
            for i in range(skip, frames):
                if rng.random() < 0.12:
                    continue                  # dropped frame
                stamps.append(round(start + i * period + rng.uniform(-0.6, 0.6) * thresh, 6))
                if rng.random() < 0.05:
                    stamps.append(round(stamps[-1] + 0.4 * period, 6))   # spurious extra frame
            if not stamps:
                stamps = [round(start, 6)]
            td[c] = sorted(stamps)
        groups, order = mod.align_time({k: list(v) for k, v in td.items()}, thresh)
        cases.append({"time_dict": td, "order_in": list(td.keys()), "thresh": thresh, "groups": groups, "cams": order})
    with open(OUT, "w") as fh:
        json.dump(cases, fh)
    print("wrote", OUT, len(cases), "cases")
    return 0


if __name__ == "__main__":
    sys.exit(main())
