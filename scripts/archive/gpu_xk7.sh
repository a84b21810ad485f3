#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export QUICK_AMD_LIB_OVERRIDE=$PWD/quick_amd/lib/libquick_amd_tools.so
KID=$(( 4 | (16 << 16) ))
for e in 64 82 336 88; do
  export QUICK_XK_ABL=$e
  bash tools/prof_passes.sh xkabl$e "--M 512 --kernel $KID --iters 24" > /dev/null 2>&1
  cp gpurun_out/pmc_xkabl$e/summary.txt gpurun_out/xk7_pmc_abl$e.txt
  rm -rf gpurun_out/pmc_xkabl$e
done
