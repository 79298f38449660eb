#!/bin/bash
# VGPR / SGPR / LDS / occupancy of every kernel of libbevwarp.so matching a pattern (default: the per-frame plan kernels):
#   bash tools/kernel_resources.sh [regex] [translation unit under csrc/, default bevwarp_plan.hip]
# Compiles to /tmp (the in-tree .so is not touched).
PAT=${1:-k_plan_}
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -Wno-pass-failed -Wno-inline-asm \
  -Rpass-analysis=kernel-resource-usage ${BEVW_CFLAGS:-} -c cameracalibration_amd/csrc/${2:-bevwarp_plan.hip} -o /tmp/libbevwarp_res.o 2>&1 |
python3 -c '
import re, sys, subprocess
pat = re.compile(sys.argv[1])
cur = None; rows = {}
for line in sys.stdin:
    m = re.search(r"remark: [^:]+:\d+:\d+: +(Function Name|Name): (\S+)", line) or re.search(r"(Function Name|Name): (\S+)", line)
    if m: cur = m.group(2); rows[cur] = {}; continue
    m = re.search(r"(VGPRs|TotalSGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\d+)", line)
    if m and cur: rows[cur][m.group(1).split()[0]] = int(m.group(2))
names = list(rows)
dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.splitlines() if names else []
for n, d in zip(names, dem):
    if pat.search(d):
        r = rows[n]
        print("%-70s vgpr %3d sgpr %3d scratch %3d lds %6d occ %d" % (d.split("(")[0][-70:], r.get("VGPRs", -1), r.get("TotalSGPRs", -1), r.get("ScratchSize", -1), r.get("LDS", -1), r.get("Occupancy", -1)))
' "$PAT"
