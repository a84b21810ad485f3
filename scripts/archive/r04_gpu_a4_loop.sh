#!/bin/bash
# timing of the parity-split (4, 2) loop (tools build, wrong results) against the (4, 2) and (4, 1) loops, same session
# (record of an experiment that was not kept: the kernel hooks it drove were removed again -- see profiles/r04_loop_variants.txt and the note in tools/gen_xw_loop.py)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export QUICK_AMD_LIB_OVERRIDE=quick_amd/lib/libquick_amd_tools.so
out=gpurun_out/r04b; mkdir -p $out
{
for shape in 512x4096x4096 512x8192x4096; do
  echo "== (4,2) S=4 $shape";           timeout 100 python tools/xk_phases.py --kernel 0x405 $shape | grep -v "reached\|word 7\|of those"
  echo "== (4,2) parity-split loop S=4 $shape (half the MFMAs)"; QUICK_XW_EXP=32 timeout 100 python tools/xk_phases.py --kernel 0x405 $shape | grep -v "reached\|word 7\|of those"
  echo "== (4,1) S=2 $shape";           timeout 100 python tools/xk_phases.py --kernel 0x1205 $shape | grep -v "reached\|word 7\|of those"
done
} > $out/a4_loop.txt 2>&1
tail -60 $out/a4_loop.txt
