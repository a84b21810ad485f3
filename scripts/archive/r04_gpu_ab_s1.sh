#!/bin/bash
# one-slice exchange-K launches: r03's library against this tree (why 3 % slower in the audit?)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
out=gpurun_out/r04c; mkdir -p $out
SH=48x4096x22016,64x4096x22016,64x4096x28672,512x11008x4096
{
for r in 1 2 3; do
  for lib in ab_r03 libquick_amd; do
    echo "== $lib (round $r)"
    QUICK_AMD_LIB_OVERRIDE=$PWD/quick_amd/lib/$lib.so timeout 300 python tools/wide_probe.py --shapes $SH --variants xk64s1=0x124 --iters 60 2>&1 | grep "us "
  done
done
} > $out/ab_s1.txt 2>&1
grep -v amdgpu $out/ab_s1.txt | cut -c1-110
