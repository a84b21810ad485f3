#!/bin/bash
# usage: tools/build_variant.sh <name> <w4a16_gemm.hip> [decode_ops.hip]  -> quick_amd/lib/ab_<name>.so (for tools/ab.sh)
cd "$(dirname "$0")/.."
name=$1; gemm=$2; dec=${3:-quick_amd/csrc/decode_ops.hip}
tmp=$(mktemp -d)
cp "$gemm" quick_amd/csrc/_ab_gemm.hip; cp "$dec" quick_amd/csrc/_ab_decode.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fno-gpu-rdc -o quick_amd/lib/ab_$name.so \
  quick_amd/csrc/_ab_gemm.hip quick_amd/csrc/repack.hip quick_amd/csrc/_ab_decode.hip
rc=$?
rm -f quick_amd/csrc/_ab_gemm.hip quick_amd/csrc/_ab_decode.hip
echo "built quick_amd/lib/ab_$name.so rc=$rc"
