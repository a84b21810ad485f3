#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export QUICK_AMD_LIB_OVERRIDE=$PWD/quick_amd/lib/libquick_amd_tools.so
(
timeout 120 python tools/xk_phases.py 512x4096x4096
for e in 80 72 88 336 320 576; do timeout 120 python tools/xk_phases.py --env-abl $e 512x4096x4096; done
for a in 18 22; do timeout 120 python tools/xk_phases.py --abl $a 512x4096x4096; done
) 2>&1 | grep -v amdgpu.ids | grep -E "abl=|K loop|first entry" | tee gpurun_out/xk6_clock.txt
