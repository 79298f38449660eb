"""Per-channel (16 TCC instances x 8 XCC) request counts of k_plan_all per placement trial from a rocprofv3 JSON
(--pmc TCC_EA0_RDREQ TCC_EA0_WRREQ --kernel-trace --output-format json around tools/r03/placement.py): is the load spread evenly over
the L2 / memory channels in a slow placement and in a fast one?   usage: channel_table.py results.json steps_per_trial"""
import json
import sys
from collections import defaultdict

import numpy as np

d = json.load(open(sys.argv[1]))["rocprofiler-sdk-tool"][0]
per = int(sys.argv[2])
kname = {k["kernel_id"]: k.get("formatted_kernel_name", k.get("kernel_name", "")) for k in d["kernel_symbols"]}
cname, inst = {}, {}
for c in d["counters"]:
    cname[c["id"]["handle"]] = c["name"]
rows = []
for rec in d["callback_records"]["counter_collection"]:
    di = rec["dispatch_data"]["dispatch_info"]
    if "k_plan_all" not in kname.get(di["kernel_id"], ""):
        continue
    dur = (rec["dispatch_data"]["end_timestamp"] - rec["dispatch_data"]["start_timestamp"]) / 1e3
    per_counter = defaultdict(list)
    for r in rec["records"]:
        per_counter[cname.get(r["counter_id"]["handle"], str(r["counter_id"]["handle"]))].append(r["value"])
    rows.append((di["dispatch_id"], dur, per_counter))
rows.sort(key=lambda t: t[0])
print("trial      us | per counter: instances, total, max/mean, min/mean, std/mean over the channel instances")
for t in range(len(rows) // per):
    chunk = rows[t * per:(t + 1) * per][3:]
    durs = sorted(c[1] for c in chunk)
    line = "%5d %7.1f |" % (t, durs[len(durs) // 2])
    for n in sorted(chunk[0][2]):
        v = np.mean([np.array(c[2][n]) for c in chunk], axis=0)
        line += " %s: n=%d tot=%.0f max/mean=%.3f min/mean=%.3f cv=%.3f |" % (n, v.size, v.sum(), v.max() / v.mean(), v.min() / v.mean(), v.std() / v.mean())
    print(line)
# one detailed table: fastest and slowest trial, RDREQ per channel (rows XCC, columns instance) when the layout is 128 values
