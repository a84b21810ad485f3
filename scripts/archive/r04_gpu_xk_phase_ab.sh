#!/bin/bash
# phase stamps of the exchange-K launches: r03's tools library against this tree's (where did the give-up protocol's 0.5 us go?)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
out=gpurun_out/r04c; mkdir -p $out
F='reached\|word 7\|of those\|amdgpu.ids'
{
for rep in 1; do
for sk in 64x11008x4096:0x824 256x4096x4096:0x444 512x11008x4096:0x244; do
  shape=${sk%:*}; k=${sk#*:}
  for lib in ab_r03_tools libquick_amd_tools ab_t_zearly ab_t_noflags ab_t_both; do
    echo "== $lib $shape $k"; QUICK_AMD_LIB_OVERRIDE=$PWD/quick_amd/lib/$lib.so timeout 100 python tools/xk_phases.py --kernel $k $shape 2>&1 | grep -v "$F"
  done
done
done
} > $out/xk_phase_ab.txt 2>&1
cat $out/xk_phase_ab.txt
