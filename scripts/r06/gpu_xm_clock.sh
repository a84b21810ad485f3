#!/bin/bash
# core clock (s_memtime over s_memrealtime, entry -> loop left) of the mid-token launch and of its timing experiments (tools builds, wrong results on purpose)
mkdir -p gpurun_out/r06
out=gpurun_out/r06/xm_clock.txt; : > $out
for v in base e48 e5 e21 e37 e53; do
  if [ $v = base ]; then export QUICK_AMD_LIB_OVERRIDE=$PWD/tools/bin/libquick_amd_tools.so; else export QUICK_AMD_LIB_OVERRIDE=$PWD/tools/bin/ab_xmt_$v.so; fi
  echo "== $v" >> $out
  timeout 300 python tools/xm_phases.py --pr 3 64x4096x22016 2>&1 | grep -v "^   phase\|landed\|end of stage" >> $out
  timeout 300 python tools/xm_phases.py --t32 --pr 1 64x4096x4096 2>&1 | grep -v "^   phase\|landed\|end of stage" >> $out
done
cat $out
