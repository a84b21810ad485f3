#!/bin/bash
# first run of the exchange-K kernels: parity against the planner's kernel + timing + phase stamps
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export QUICK_AMD_LIB_OVERRIDE=$PWD/quick_amd/lib/libquick_amd_tools.so
XK=4
v() { echo $(( XK | ($1 << 4) | ($2 << 8) | ($3 << 22) | ($4 << 26) )); }
V="auto=0,xk=$(v 0 0 0 0),xk_s1=$(v 4 1 0 0),xk_s2=$(v 4 2 0 0),xk_n4w4=$(v 4 0 4 4),xk_n4w5=$(v 4 0 4 5),xk_n5w5=$(v 4 0 5 5),xk_n5w6=$(v 4 0 5 6)"
timeout 900 python tools/wide_probe.py --shapes 512x4096x4096,300x4096x4096,256x4096x4096,128x4096x4096,512x4096x12288,512x11008x4096,1024x4096x4096 \
   --variants "$V" --out gpurun_out/xk1_probe.jsonl 2>&1 | tee gpurun_out/xk1_probe.txt
V2="auto=0,xk2=$(v 2 0 0 0),xk2_s4=$(v 2 4 0 0),xk2_s2=$(v 2 2 0 0),xk4_s8=$(v 4 8 0 0),xk2_n4=$(v 2 0 4 4)"
timeout 600 python tools/wide_probe.py --shapes 64x4096x4096,128x4096x4096,64x11008x4096,48x4096x4096,64x4096x12288 \
   --variants "$V2" --out gpurun_out/xk1_probe2.jsonl 2>&1 | tee gpurun_out/xk1_probe2.txt
timeout 300 python tools/xk_phases.py 512x4096x4096 256x4096x4096 2>&1 | tee gpurun_out/xk1_phases.txt
timeout 300 python tools/xk_phases.py --kernel $(v 2 8 0 0) 64x4096x4096 2>&1 | tee -a gpurun_out/xk1_phases.txt
unset QUICK_AMD_LIB_OVERRIDE
timeout 900 python -m pytest tests -m gpu -x -q -k "baseline or golden or pin or way_out" 2>&1 | tail -5 | tee gpurun_out/xk1_tests.txt
