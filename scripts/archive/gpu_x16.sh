#!/bin/bash
# the sixteen-wave 64 x 128 exchange-K tile (kernel bit 13) against the eight-wave one
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export QUICK_AMD_LIB_OVERRIDE=$PWD/quick_amd/lib/libquick_amd_tools.so
XK=4
v() { echo $(( XK | ($1 << 4) | ($2 << 8) | $3 )); }
W16=$((1<<13))
timeout 600 python tools/wide_probe.py --shapes ${SHAPES:-512x4096x4096,300x4096x4096,448x4096x4096,512x11008x4096,512x8192x4096,200x4096x4096,64x4096x22016,64x4096x28672,77x1024x768} \
   --variants "auto=0,xk2=$(v 2 1 0),x16=$(v 2 1 $W16),xk4=$(v 4 0 0)" --iters 30 2>&1 | grep -v amdgpu.ids | cut -c1-150 | tee gpurun_out/x16_probe.txt
(
timeout 120 python tools/xk_phases.py --kernel $(v 2 1 $W16) 512x4096x4096
timeout 120 python tools/xk_phases.py --kernel $(v 2 1 0) 512x4096x4096
) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/x16_phases.txt
timeout 120 python tools/xk_phases.py --kernel $(v 2 1 $W16) --env-abl 262208 512x4096x4096 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/x16_phases.txt
