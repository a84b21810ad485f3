#!/bin/bash
# r04 A/B in one session: r03's library (exchange-K kernels that spin and trap) against this tree's (give-up protocol), forced xk kernels
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04
SH=128x4096x4096,96x11008x4096,64x11008x4096,64x14336x4096,64x4096x6144,64x4096x12288,256x4096x4096,512x11008x4096,48x4096x6144
V=xk64=0x24,xk128=0x44
{
for r in 1 2; do
  echo "== r03 library (round $r)"; QUICK_AMD_LIB_OVERRIDE=$PWD/quick_amd/lib/ab_r03.so timeout 300 python tools/wide_probe.py --shapes $SH --variants $V --iters 60
  echo "== this tree (round $r)"; timeout 300 python tools/wide_probe.py --shapes $SH --variants $V --iters 60
done
} > gpurun_out/r04/ab_xk.txt 2>&1
grep -v amdgpu.ids gpurun_out/r04/ab_xk.txt | cut -c1-150
