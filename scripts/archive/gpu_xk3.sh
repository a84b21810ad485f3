#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export QUICK_AMD_LIB_OVERRIDE=$PWD/quick_amd/lib/libquick_amd_tools.so
XK=4
v() { echo $(( XK | ($1 << 4) | ($2 << 8) | ($3 << 22) | ($4 << 26) )); }
V="auto=0,xk=$(v 0 0 0 0),xk_s1=$(v 4 1 0 0),xk_s2=$(v 4 2 0 0),xk_s4=$(v 4 4 0 0),xk2=$(v 2 0 0 0),xk2_s4=$(v 2 4 0 0),xk2_s2=$(v 2 2 0 0),xk2_s1=$(v 2 1 0 0)"
timeout 900 python tools/wide_probe.py --shapes 512x4096x4096,300x4096x4096,256x4096x4096,128x4096x4096,64x4096x4096,33x4096x4096,512x11008x4096,64x11008x4096,200x8192x1024,64x4096x12288,17x2048x2048 \
   --variants "$V" --out gpurun_out/xk3_probe.jsonl 2>&1 | grep -v amdgpu.ids | tee gpurun_out/xk3_probe.txt
(
timeout 120 python tools/xk_phases.py 512x4096x4096
for e in 192 320 576 1088; do timeout 120 python tools/xk_phases.py --env-abl $e 512x4096x4096; done
timeout 120 python tools/xk_phases.py --kernel $(v 2 8 0 0) 64x4096x4096
timeout 120 python tools/xk_phases.py --kernel $(v 4 4 0 0) 256x4096x4096
) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/xk3_phases.txt
timeout 300 python tools/dense_ref.py 512x4096x4096 64x4096x4096 4096x4096x4096 2>&1 | grep -v amdgpu.ids | tee gpurun_out/xk3_dense.txt
