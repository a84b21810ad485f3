#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export QUICK_AMD_LIB_OVERRIDE=$PWD/quick_amd/lib/libquick_amd_tools.so
XK=4; LD=$((1<<12))
v() { echo $(( XK | ($1 << 4) | ($2 << 8) | $3 )); }
V="auto=0,xk2=$(v 2 0 0),xl2=$(v 2 0 $LD),xk4=$(v 4 0 0),xl4=$(v 4 0 $LD),xl4_s1=$(v 4 1 $LD),xl2_s2=$(v 2 2 $LD)"
timeout 900 python tools/wide_probe.py --shapes 512x4096x4096,300x4096x4096,256x4096x4096,128x4096x4096,64x4096x4096,33x1024x512,512x11008x4096,64x11008x4096,1024x4096x4096,200x8192x1024,130x4096x256 \
   --variants "$V" --iters 30 --out gpurun_out/xl1_probe.jsonl 2>&1 | grep -v amdgpu.ids | cut -c1-175 | tee gpurun_out/xl1_probe.txt
(
timeout 120 python tools/xk_phases.py --kernel $(v 2 1 $LD) 512x4096x4096
timeout 120 python tools/xk_phases.py --kernel $(v 4 2 $LD) 512x4096x4096
timeout 120 python tools/xk_phases.py --kernel $(v 2 1 0) 512x4096x4096
timeout 120 python tools/xk_phases.py --kernel $(v 4 2 0) 512x4096x4096
) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/xl1_phases.txt
