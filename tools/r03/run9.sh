#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_run9
mkdir -p $O
ls /sys/kernel/debug 2>&1 | head -5
ls /sys/kernel/debug/dri 2>&1 | head
mount | grep -i debug | head -3
ls /sys/class/drm/ | head; ls /sys/class/drm/card*/device/ 2>/dev/null | grep -i -E "mem|vram|frag|part" | head -20
cat /sys/module/amdgpu/parameters/vm_fragment_size /sys/module/amdgpu/parameters/vm_block_size /sys/module/amdgpu/parameters/vram_limit 2>&1 | head
uname -r; cat /sys/module/amdgpu/version 2>/dev/null
