#!/bin/bash
# A/B aid: tools/bin/ab_<name>.so = the product library with extra flags on chosen translation units
#   bash tools/build_variant.sh <name> "-DQA_EXP_..." ["w4a16_xk w4a16_xw"] [tools]
# (third argument: the units to recompile with the flags, default the exchange kernels; the lean units keep their kernarg preload)
name=$1; defs=$2; units=${3:-"w4a16_xk w4a16_xw"}
root=$(cd "$(dirname "$0")/.." && pwd)
base=$root/quick_amd/lib/obj
if [ "$4" = tools ]; then base=$root/tools/bin/obj_tools; defs="$defs -DQUICK_AMD_TOOLS"; python -m quick_amd.build --tools > /dev/null || exit 1; else python -m quick_amd.build > /dev/null || exit 1; fi
o=$root/tools/bin/obj_ab_$name; mkdir -p $o
all="w4a16_gemm w4a16_xk w4a16_xw w4a16_xm w4a16_lean w4a16_lean_a w4a16_lean_b w4a16_lean_c repack decode_ops"
objs=""
for f in $all; do
  if echo " $units " | grep -q " $f "; then
    extra=""; case $f in w4a16_lean_*|w4a16_xm|w4a16_xw|w4a16_xk) extra="-mllvm -amdgpu-kernarg-preload-count=16";; esac
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc $extra $defs -c -o $o/$f.o $root/quick_amd/csrc/$f.hip &
    objs="$objs $o/$f.o"
  else
    objs="$objs $base/$f.o"
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fno-gpu-rdc -o $root/tools/bin/ab_$name.so $objs
ls -la $root/tools/bin/ab_$name.so
