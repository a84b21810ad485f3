#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export QUICK_AMD_LIB_OVERRIDE=$PWD/quick_amd/lib/libquick_amd_tools.so
B="python bench.py --steps 20 --warmup 5 --sweep 512 --layers= --prefill-layers= --cpu-seconds 0 --decode-seconds 0"
K=$((4+(2<<4)+(1<<8)+(16<<16)))
for rep in 1 2 3 4; do
  for e in 0 32768; do
    QUICK_XK_ABL=$e $B --kernel $K 2>&1 >/dev/null | grep "M= 512" | sed "s/^/abl=$e /"
  done
done | tee gpurun_out/r03_ab_wt.txt
timeout 120 python tools/xk_phases.py --kernel $((4+(2<<4)+(1<<8))) 512x4096x4096 2>&1 | grep -v amdgpu | tee gpurun_out/r03_phases_m512_xk2.txt
