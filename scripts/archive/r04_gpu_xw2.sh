#!/bin/bash
# r04: 128 x 256 four-wave kernels, second build: parity (also with every wave giving its block up), timing, phase stamps
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04
{
timeout 300 python tools/xw_check.py
echo "== every wave gives up at once (poll limit 2 ticks)"
QUICK_AMD_EXCHANGE_POLL_LOG2=1 timeout 300 python tools/xw_check.py 128x512x256 300x2048x512 512x4096x4096 77x4096x256
echo "== poll limit 2.5 us"
QUICK_AMD_EXCHANGE_POLL_LOG2=8 timeout 300 python tools/xw_check.py 300x2048x512 512x4096x4096 512x4096x4096
echo "== timing"
timeout 300 python tools/wide_probe.py --shapes 512x4096x4096,1024x4096x4096,256x4096x4096,512x8192x8192 --variants auto=0,xw4=0x405,xw2=0x205,xw1=0x105 --iters 60
export QUICK_AMD_LIB_OVERRIDE=$PWD/quick_amd/lib/libquick_amd_tools.so
echo "== phases"
timeout 200 python tools/xk_phases.py --kernel 0x405 512x4096x4096
timeout 200 python tools/xk_phases.py --kernel 0x405 --abl 20 512x4096x4096
timeout 200 python tools/xk_phases.py --kernel 0x405 --abl 21 512x4096x4096
timeout 200 python tools/xk_phases.py --kernel 0x205 1024x4096x4096
} > gpurun_out/r04/xw2.txt 2>&1
grep -v "amdgpu.ids" gpurun_out/r04/xw2.txt | tail -90
