#!/bin/bash
# what would fewer barriers in the K loop buy?  timing builds (results may be wrong): a barrier behind every second / fourth stage only
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export QUICK_AMD_LIB_OVERRIDE=$PWD/quick_amd/lib/libquick_amd_tools.so
XK=4
v() { echo $(( XK | ($1 << 4) | ($2 << 8) )); }
(
timeout 120 python tools/xk_phases.py --kernel $(v 2 1) 512x4096x4096
timeout 120 python tools/xk_phases.py --kernel $(v 2 1) --env-abl 524352 512x4096x4096
timeout 120 python tools/xk_phases.py --kernel $(v 2 1) --env-abl 1572928 512x4096x4096
timeout 120 python tools/xk_phases.py --kernel $(v 4 2) --env-abl 524352 512x4096x4096
) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/bar.txt
