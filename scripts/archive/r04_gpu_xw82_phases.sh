#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export QUICK_AMD_LIB_OVERRIDE=quick_amd/lib/libquick_amd_tools.so
out=gpurun_out/r04b; mkdir -p $out
F='reached\|word 7\|of those\|amdgpu.ids'
{
for shape in 4096x4096x4096 8192x4096x4096; do
echo "== xw 256 x 256 $shape"; timeout 100 python tools/xk_phases.py --kernel 0x85 $shape 2>&1 | grep -v "$F"
echo "== xw 128 x 256 S=1 $shape"; timeout 100 python tools/xk_phases.py --kernel 0x105 $shape 2>&1 | grep -v "$F"
done
} > $out/xw82_phases.txt 2>&1
cat $out/xw82_phases.txt
