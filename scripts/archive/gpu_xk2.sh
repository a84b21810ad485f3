#!/bin/bash
# exchange-K kernels, second run: parity after the mailbox fix, anatomy (phase stamps x ablation builds), PMC passes
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export QUICK_AMD_LIB_OVERRIDE=$PWD/quick_amd/lib/libquick_amd_tools.so
XK=4
v() { echo $(( XK | ($1 << 4) | ($2 << 8) | ($3 << 22) | ($4 << 26) )); }
V="auto=0,xk=$(v 0 0 0 0),xk_s1=$(v 4 1 0 0),xk_s2=$(v 4 2 0 0),xk_s4=$(v 4 4 0 0),xk2=$(v 2 0 0 0),xk2_s4=$(v 2 4 0 0),xk2_s2=$(v 2 2 0 0)"
timeout 900 python tools/wide_probe.py --shapes 512x4096x4096,300x4096x4096,256x4096x4096,128x4096x4096,64x4096x4096,33x4096x4096,512x11008x4096,64x11008x4096,200x8192x1024 \
   --variants "$V" --out gpurun_out/xk2_probe.jsonl 2>&1 | tee gpurun_out/xk2_probe.txt
for abl in 16 17 18 19 20 21 22 23 24; do
  timeout 120 python tools/xk_phases.py --abl $abl 512x4096x4096 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/xk2_phases.txt
bash tools/prof_passes.sh xk512 "--M 512 --kernel 4" > /dev/null 2>&1
cp gpurun_out/pmc_xk512/summary.txt gpurun_out/xk2_pmc_m512.txt; cat gpurun_out/xk2_pmc_m512.txt
