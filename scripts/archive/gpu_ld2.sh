#!/bin/bash
cd "$(dirname "$0")/.."
export QUICK_AMD_LIB_OVERRIDE=$PWD/quick_amd/lib/libquick_amd_tools.so
(
for e in 64 4160 4672 320 576; do timeout 120 python tools/xk_phases.py --env-abl $e 512x4096x4096; done
timeout 120 python tools/xk_phases.py --abl 18 512x4096x4096
) 2>&1 | grep -v amdgpu.ids | grep -E "abl=|K loop:" | sed 's/: xk tokens.*//' | paste - - | tee gpurun_out/ld2.txt
