#!/bin/bash
# 64-token exchange-K tiles, K slices as co-resident PAIRS of workgroups per CU (four waves per SIMD)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export QUICK_AMD_LIB_OVERRIDE=$PWD/quick_amd/lib/libquick_amd_tools.so
XK=4
v() { echo $(( XK | ($1 << 4) | ($2 << 8) )); }
timeout 600 python tools/wide_probe.py --shapes 512x4096x4096,256x4096x4096,384x4096x4096,512x11008x4096,512x4096x12288,256x4096x12288 \
   --variants "auto=0,xk2s1=$(v 2 1),xk2s2=$(v 2 2),xk2s4=$(v 2 4),xk4=$(v 4 0)" --iters 30 2>&1 | grep -v amdgpu.ids | cut -c1-150 | tee gpurun_out/occ2_probe.txt
(
timeout 120 python tools/xk_phases.py --kernel $(v 2 2) 512x4096x4096
) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/occ2_phases.txt
