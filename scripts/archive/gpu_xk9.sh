#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export QUICK_AMD_LIB_OVERRIDE=$PWD/quick_amd/lib/libquick_amd_tools.so
(
for e in 336 16720; do timeout 120 python tools/xk_phases.py --env-abl $e 512x4096x4096; done
timeout 120 python tools/xk_phases.py --abl 19 512x4096x4096
) 2>&1 | grep -v amdgpu.ids | grep -E "abl=|K loop|first entry|clocks" | tee gpurun_out/xk9.txt
