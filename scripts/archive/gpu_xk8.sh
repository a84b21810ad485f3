#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export QUICK_AMD_LIB_OVERRIDE=$PWD/quick_amd/lib/libquick_amd_tools.so
(
for e in 64 8256 8258; do timeout 120 python tools/xk_phases.py --env-abl $e 512x4096x4096; done
) 2>&1 | grep -v amdgpu.ids | grep -E "abl=|K loop|first entry|clocks" | tee gpurun_out/xk8_wait.txt
