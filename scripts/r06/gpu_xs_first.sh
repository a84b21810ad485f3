#!/bin/bash
# first run of the split-role mid-token kernels: results against the oracle, then timing against the eight-wave kernels
mkdir -p gpurun_out/r06
out=gpurun_out/r06/xs_first.txt; : > $out
timeout 600 python tools/xm_check.py --no-time --only-xs 64x1024x352 50x1536x4096 64x4096x4096 33x2048x1024 >> $out 2>&1
echo "rc $?" >> $out
timeout 600 python tools/xm_check.py --no-check --only-xm --only-xs 64x4096x12288 64x4096x22016 48x4096x22016 64x4096x14336 64x8192x10240 >> $out 2>&1
echo "rc $?" >> $out
cat $out
