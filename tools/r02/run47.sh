#!/bin/bash
# what bounds k_lum_groups / k_vsum / k_gain_lut (config 4)?  utilisation counters per kernel
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
i=0
for set in "GRBM_GUI_ACTIVE TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum" "TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum"; do
  i=$((i+1)); rm -rf /tmp/pq
  timeout 90 rocprofv3 --pmc $set --output-format csv -d /tmp/pq -- python $R/bench.py --workload blend_balance_b256 --steps 3 --warmup 1 --no-cpu-baseline > /tmp/pq.log 2>&1
  cp $(find /tmp/pq -name "*counter_collection.csv" | head -1) /tmp/pass_$i.csv
done
python - <<'PY'
import csv, glob
from collections import defaultdict
tot = defaultdict(lambda: defaultdict(float)); n = defaultdict(lambda: defaultdict(int))
for f in sorted(glob.glob("/tmp/pass_*.csv")):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        for key in ("k_lum_groups", "k_vsum", "k_gain_lut", "k_plan_all"):
            if key in k:
                tot[key][r["Counter_Name"]] += float(r["Counter_Value"]); n[key][r["Counter_Name"]] += 1
for k in tot:
    v = {c: tot[k][c] / n[k][c] for c in tot[k]}
    g = v["GRBM_GUI_ACTIVE"] / 8
    print("%s: %.0f us; TA busy %.0f%%, L1 waiting %.0f%%, VALU %.0f%%, LDS %.0f%% (conflicts %.0f%% of LDS cycles), wait_inst %.0f%% (lds %.0f%%); TCP rd %.2f M wr %.2f M -> %.1f G req/s; EA rd %.2f M wr %.2f M; VALU insts %.1f M" % (
        k, g / 2.4e3, 100 * v["TA_TA_BUSY_sum"] / 256 / g, 100 * v["TCP_PENDING_STALL_CYCLES_sum"] / 256 / g, 100 * v["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / g,
        100 * v["SQ_ACTIVE_INST_LDS"] * 4 / 1024 / g, 100 * v["SQ_LDS_BANK_CONFLICT"] / max(1, v["SQ_LDS_IDX_ACTIVE"]), 100 * v["SQ_WAIT_INST_ANY"] / v["SQ_WAVE_CYCLES"],
        100 * v["SQ_WAIT_INST_LDS"] / v["SQ_WAVE_CYCLES"], v["TCP_TCC_READ_REQ_sum"] / 1e6, v["TCP_TCC_WRITE_REQ_sum"] / 1e6,
        (v["TCP_TCC_READ_REQ_sum"] + v["TCP_TCC_WRITE_REQ_sum"]) / (g / 2.4e9) / 1e9, v["TCC_EA0_RDREQ_sum"] / 1e6, v["TCC_EA0_WRREQ_sum"] / 1e6, v["SQ_INSTS_VALU"] / 1e6))
PY
