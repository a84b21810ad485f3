#!/bin/bash
# timing experiments on the split-role loops (tools/bin/ab_xs_e<n>.so from XS_EXP=<n> tools/gen_xm_loop.py; wrong results on purpose), in-kernel span
mkdir -p gpurun_out/r06
out=gpurun_out/r06/xs_exp.txt; : > $out
for r in 1 2; do
for e in base ${XS_EXPS:-1 2 3 4 8 12 16 48}; do
  if [ $e = base ]; then unset QUICK_AMD_LIB_OVERRIDE; else export QUICK_AMD_LIB_OVERRIDE=$PWD/tools/bin/ab_xs_e$e.so; fi
  echo "== exp $e (round $r)" >> $out
  timeout 300 python tools/xm_check.py --no-check --only-xm --only-xs ${XS_SHAPES:-64x4096x12288 64x4096x22016} 2>&1 | grep "   " | grep "xs\|xm pr=3  \|xm pr=2  " | cut -c1-75 >> $out
done; done
cat $out
