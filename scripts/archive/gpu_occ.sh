#!/bin/bash
# does a 64 x 128 exchange-K tile run faster with FOUR waves per SIMD?  Two workgroups per CU (<= 128 registers, tools build) on shapes with 512 tiles
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export QUICK_AMD_LIB_OVERRIDE=$PWD/quick_amd/lib/libquick_amd_tools.so
XK=4
k2=$(( XK | (2 << 4) | (1 << 8) ))
k2n4=$(( k2 | (4 << 22) ))
(
for sh in 1024x4096x4096 512x4096x8192 512x4096x4096; do
  timeout 120 python tools/xk_phases.py --kernel $k2 $sh
  timeout 120 python tools/xk_phases.py --kernel $k2 --env-abl 131136 $sh
  timeout 120 python tools/xk_phases.py --kernel $k2n4 $sh
  timeout 120 python tools/xk_phases.py --kernel $k2n4 --env-abl 131136 $sh
done
) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/occ.txt
