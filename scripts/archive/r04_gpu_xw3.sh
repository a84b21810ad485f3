#!/bin/bash
# r04: the three tile shapes of the four-wave kernels: parity (also with every wave giving its block up), timing, phase stamps
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04
{
timeout 400 python tools/xw_check.py
echo "== every wave gives up at once (poll limit 2 ticks)"
QUICK_AMD_EXCHANGE_POLL_LOG2=1 timeout 300 python tools/xw_check.py 128x512x256 300x2048x512 512x4096x4096 77x4096x256
echo "== timing"
timeout 300 python tools/wide_probe.py --shapes 512x4096x4096,1024x4096x4096,256x4096x4096,128x4096x4096,64x4096x4096,64x4096x11008,64x4096x12288,512x4096x11008 --variants auto=0,xw42=0x5,xw41=0x1005,xw21=0x25,xw41s1=0x1105,xw21s1=0x125 --iters 60
export QUICK_AMD_LIB_OVERRIDE=$PWD/quick_amd/lib/libquick_amd_tools.so
echo "== phases"
timeout 200 python tools/xk_phases.py --kernel 0x125 512x4096x4096
timeout 200 python tools/xk_phases.py --kernel 0x1205 512x4096x4096
timeout 200 python tools/xk_phases.py --kernel 0x405 512x4096x4096
} > gpurun_out/r04/xw3.txt 2>&1
grep -v "amdgpu.ids" gpurun_out/r04/xw3.txt | grep -v " ok   \[" | tail -100
