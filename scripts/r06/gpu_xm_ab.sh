#!/bin/bash
# A/B of mid-token loop variants (tools/bin/ab_xm_<name>.so, right results), alternating
mkdir -p gpurun_out/r06
out=gpurun_out/r06/xm_ab_${XM_TAG:-1}.txt; : > $out
for r in 1 2; do
for n in $XM_VARIANTS; do
  if [ $n = base ]; then unset QUICK_AMD_LIB_OVERRIDE; else export QUICK_AMD_LIB_OVERRIDE=$PWD/tools/bin/ab_xm_$n.so; fi
  echo "== $n (round $r)" >> $out
  timeout 300 python tools/xm_check.py --no-check --only-xm ${XM_SHAPES:-64x4096x4096 64x4096x12288 64x4096x22016 64x11008x4096} 2>&1 | grep "   " | grep -v big | cut -c1-100 >> $out
done; done
cat $out
