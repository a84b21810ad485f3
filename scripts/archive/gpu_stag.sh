#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export QUICK_AMD_LIB_OVERRIDE=$PWD/quick_amd/lib/libquick_amd_tools.so
XK=4
v() { echo $(( XK | ($1 << 4) | ($2 << 8) )); }
(
for rep in 1 2; do
timeout 120 python tools/xk_phases.py --kernel $(v 2 1) 512x4096x4096
timeout 120 python tools/xk_phases.py --kernel $(v 2 1) --env-abl 65600 512x4096x4096
timeout 120 python tools/xk_phases.py --kernel $(v 4 2) 512x4096x4096
timeout 120 python tools/xk_phases.py --kernel $(v 4 2) --env-abl 65600 512x4096x4096
done
) 2>&1 | grep -v amdgpu.ids | grep -E "abl=|K loop|first entry" | tee gpurun_out/stag.txt
