#!/bin/bash
# early barrier (one counted wait per stage, a stage more for every load to arrive) against the r04 loops, same session; tools library
# (record of an experiment that was not kept: the kernel hooks it drove were removed again -- see profiles/r04_loop_variants.txt and the note in tools/gen_xw_loop.py)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export QUICK_AMD_LIB_OVERRIDE=quick_amd/lib/libquick_amd_tools.so
out=gpurun_out/r04b; mkdir -p $out
{
echo "## correctness of the early-barrier loops (stamped builds)"
QUICK_XW_EXP=64 XW_EXTRA_BITS=0x10000 timeout 300 python tools/xw_check.py 2>&1 | grep -v amdgpu.ids
F='reached\|word 7\|of those\|amdgpu.ids'
for shape in 512x4096x4096 512x8192x4096; do
  for e in 0 64 16; do
    echo "== (4,1) S=2 $shape experiment $e";  QUICK_XW_EXP=$e timeout 100 python tools/xk_phases.py --kernel 0x1205 $shape 2>&1 | grep -v "$F"
  done
  for e in 0 64; do
    echo "== (4,2) S=4 $shape experiment $e";  QUICK_XW_EXP=$e timeout 100 python tools/xk_phases.py --kernel 0x405 $shape 2>&1 | grep -v "$F"
  done
  for e in 0 64 16; do
    echo "== (2,1) S=1 $shape experiment $e";  QUICK_XW_EXP=$e timeout 100 python tools/xk_phases.py --kernel 0x125 $shape 2>&1 | grep -v "$F"
  done
done
} > $out/bar.txt 2>&1
grep -v "entry ->\|way out\|slices exch" $out/bar.txt | tail -90
