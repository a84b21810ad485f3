#!/bin/bash
# correctness of the product loop, then timing experiments on the mid-token loop (wrong results on purpose): tools/bin/ab_xm_e<n>.so built by
# tools/build_variant.sh from XM_EXP=<n> loops
mkdir -p gpurun_out/r06
timeout 600 python tools/xm_check.py --no-time 64x4096x4096 33x4096x4096 17x1024x256 50x1536x4096 40x11008x512 > gpurun_out/r06/xm_check2.txt 2>&1
grep -c "WRONG" gpurun_out/r06/xm_check2.txt; tail -3 gpurun_out/r06/xm_check2.txt
out=gpurun_out/r06/xm_exp2.txt; : > $out
for e in base ${XM_EXPS:-16 32 48}; do
  if [ $e = base ]; then unset QUICK_AMD_LIB_OVERRIDE; else export QUICK_AMD_LIB_OVERRIDE=$PWD/tools/bin/ab_xm_e$e.so; fi
  echo "== exp $e" >> $out
  timeout 300 python tools/xm_check.py --no-check ${XM_ONLY---only-xm} ${XM_SHAPES:-64x4096x4096 64x4096x12288 64x4096x22016 64x11008x4096 24x4096x4096} 2>&1 | grep "   " >> $out
done
cut -c1-132 $out
