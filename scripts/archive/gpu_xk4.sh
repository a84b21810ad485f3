#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export QUICK_AMD_LIB_OVERRIDE=$PWD/quick_amd/lib/libquick_amd_tools.so
(
for rep in 1 2; do
timeout 120 python tools/xk_phases.py 512x4096x4096
for e in 4160 320 576 1088; do timeout 120 python tools/xk_phases.py --env-abl $e 512x4096x4096; done
done
) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/xk4_phases.txt
