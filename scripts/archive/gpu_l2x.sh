#!/bin/bash
# exchange-K: the slices of a tile on one XCD, mailboxes polled in that XCD's L2 first (kernel bit 21)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export QUICK_AMD_LIB_OVERRIDE=$PWD/quick_amd/lib/libquick_amd_tools.so
XK=4; L2=$((1<<21))
v() { echo $(( XK | ($1 << 4) | ($2 << 8) | $3 )); }
timeout 600 python tools/wide_probe.py --shapes ${SHAPES:-64x4096x4096,128x4096x4096,256x4096x4096,512x4096x4096,64x11008x4096,64x4096x12288,64x4096x6144,512x11008x4096,160x8192x8192} \
   --variants "auto=0,xk2=$(v 2 0 0),xk2l=$(v 2 0 $L2),xk4=$(v 4 0 0),xk4l=$(v 4 0 $L2)" --iters 30 2>&1 | grep -v amdgpu.ids | cut -c1-170 | tee gpurun_out/l2x_probe.txt
(
timeout 120 python tools/xk_phases.py --kernel $(v 2 8 0) 64x4096x4096
timeout 120 python tools/xk_phases.py --kernel $(v 2 8 $L2) 64x4096x4096
timeout 120 python tools/xk_phases.py --kernel $(v 4 2 0) 512x4096x4096
timeout 120 python tools/xk_phases.py --kernel $(v 4 2 $L2) 512x4096x4096
) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/l2x_phases.txt
