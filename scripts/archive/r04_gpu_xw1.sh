#!/bin/bash
# r04: first run of the 128 x 256 four-wave kernels: parity, timing against the planner's pick, phase stamps
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04
{
timeout 300 python tools/xw_check.py
echo "== timing"
timeout 300 python tools/wide_probe.py --shapes 512x4096x4096,1024x4096x4096,256x4096x4096 --variants auto=0,xw4=0x405,xw2=0x205,xw1=0x105 --iters 60
export QUICK_AMD_LIB_OVERRIDE=$PWD/quick_amd/lib/libquick_amd_tools.so
echo "== phases"
timeout 200 python tools/xk_phases.py --kernel 0x405 512x4096x4096
timeout 200 python tools/xk_phases.py --kernel 0x405 --abl 20 512x4096x4096
timeout 200 python tools/xk_phases.py --kernel 0x205 1024x4096x4096
} > gpurun_out/r04/xw1.txt 2>&1
tail -80 gpurun_out/r04/xw1.txt
