#!/bin/bash
# the split-role mid-token kernels (tools/patches/r06_xs_split_roles.patch applied): results against the oracle, then timing against the eight-wave kernels
mkdir -p gpurun_out/r06
out=gpurun_out/r06/xs_first.txt; : > $out
timeout 600 python tools/xm_check.py --no-time --only-xs 50x1536x4096 64x4096x4096 33x2048x1024 17x1024x256 >> $out 2>&1
echo "rc $?" >> $out
timeout 900 python tools/xm_check.py --no-check --only-xm --only-xs ${XS_SHAPES:-64x4096x4096 48x4096x4096 32x4096x4096 64x8192x8192 64x4096x12288 64x4096x22016} 2>&1 | cut -c1-150 >> $out
echo "rc $?" >> $out
cat $out
