#!/bin/bash
# rows stored straight from the registers (no LDS image, no barrier) against the image path, 128 x 128 tiles, two slices; tools library
# (record of an experiment that was not kept: the kernel hooks it drove were removed again -- see profiles/r04_loop_variants.txt and the note in tools/gen_xw_loop.py)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export QUICK_AMD_LIB_OVERRIDE=quick_amd/lib/libquick_amd_tools.so
out=gpurun_out/r04b; mkdir -p $out
{
echo "## correctness (stamped build with direct rows)"
XW_EXTRA_BITS=0x110000 timeout 300 python tools/xw_check.py 512x4096x4096 300x2048x512 130x1024x512 2>&1 | grep -v amdgpu.ids
F='reached\|word 7\|of those\|amdgpu.ids'
for rep in 1 2 3; do
for shape in 512x4096x4096; do
  echo "== image $shape";  timeout 100 python tools/xk_phases.py --kernel 0x1205 $shape 2>&1 | grep -v "$F"
  echo "== direct $shape"; timeout 100 python tools/xk_phases.py --kernel 0x1205 --abl 17 $shape 2>&1 | grep -v "$F"
done
done
} > $out/direct.txt 2>&1
cat $out/direct.txt | tail -80
