"""Join rocprofv3 counter rows and kernel durations of k_plan_all per placement trial (tools/r03/placement.py runs 3 warm-up + K timed
steps per trial).  usage: placement_table.py counters.csv trace.csv steps_per_trial"""
import csv
import sys
from collections import defaultdict

cfile, tfile, per = sys.argv[1], sys.argv[2], int(sys.argv[3])
dur = {}
for r in csv.DictReader(open(tfile)):
    if "k_plan_all" in r["Kernel_Name"]:
        dur[int(r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
vals = defaultdict(dict)
for r in csv.DictReader(open(cfile)):
    if "k_plan_all" in r["Kernel_Name"]:
        vals[int(r["Dispatch_Id"])][r["Counter_Name"]] = vals[int(r["Dispatch_Id"])].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
ids = sorted(vals)
names = sorted({c for d in vals.values() for c in d})
print("%5s %9s " % ("trial", "us") + " ".join("%26s" % n[-26:] for n in names))
for t in range(len(ids) // per):
    chunk = ids[t * per:(t + 1) * per][3:]           # skip the warm-up dispatches of the trial
    d = sorted(dur.get(i, 0.0) for i in chunk)
    med = d[len(d) // 2] if d else 0.0
    row = ["%26.0f" % (sum(vals[i].get(n, 0.0) for i in chunk) / max(1, len(chunk))) for n in names]
    print("%5d %9.1f " % (t, med) + " ".join(row))
