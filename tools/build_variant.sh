#!/bin/bash
# A/B aid: quick_amd/lib/ab_<name>.so = the product library with extra -D flags on the exchange kernels' translation units
#   bash tools/build_variant.sh <name> "-DQA_EXP_..." [tools]
name=$1; defs=$2; base=obj
root=$(cd "$(dirname "$0")/.." && pwd)
if [ "$3" = tools ]; then base=obj_tools; defs="$defs -DQUICK_AMD_TOOLS"; python -m quick_amd.build --tools > /dev/null || exit 1; else python -m quick_amd.build > /dev/null || exit 1; fi
o=$root/quick_amd/lib/obj_ab_$name; mkdir -p $o
for f in w4a16_xk w4a16_xw; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc $defs -c -o $o/$f.o $root/quick_amd/csrc/$f.hip &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fno-gpu-rdc -o $root/quick_amd/lib/ab_$name.so $root/quick_amd/lib/$base/w4a16_gemm.o $o/w4a16_xk.o $o/w4a16_xw.o $root/quick_amd/lib/$base/repack.o $root/quick_amd/lib/$base/decode_ops.o
ls -la $root/quick_amd/lib/ab_$name.so
