#!/bin/bash
# r04: anatomy of the 64 x 128 and 128 x 128 loops (tools builds: experiments on the generated loop), the two-slice exchange with one poll
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04
export QUICK_AMD_LIB_OVERRIDE=$PWD/quick_amd/lib/libquick_amd_tools.so
{
for e in 0 1 2 4 8 15; do echo "== 64 x 128, experiment $e"; QUICK_XW_EXP=$e timeout 100 python tools/xk_phases.py --kernel 0x125 512x4096x4096 | grep -v "reached\|word 7\|of those"; done
for e in 0 1 2 4 8; do echo "== 128 x 128 S = 2, experiment $e"; QUICK_XW_EXP=$e timeout 100 python tools/xk_phases.py --kernel 0x1205 512x4096x4096 | grep -v "reached\|word 7\|of those"; done
unset QUICK_AMD_LIB_OVERRIDE
timeout 200 python tools/xw_check.py 512x4096x4096 300x2048x512
timeout 300 python tools/wide_probe.py --shapes 512x4096x4096,256x4096x4096 --variants auto=0,xw41=0x1005,xw21=0x25 --iters 60
} > gpurun_out/r04/xw4.txt 2>&1
grep -v "amdgpu.ids" gpurun_out/r04/xw4.txt | grep -v " ok   \[" | tail -120
