#!/bin/bash
# 17..64 tokens on the model layers: the planner's pick against forced skinny flavours, the tiled kernel and the exchange-K tile
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
SK=1
s() { echo $(( SK | ($1 << 4) | ($2 << 8) | $3 )); }
X=$((1<<12))
V="warm=0,auto=0,n4w8=$(s 4 2 0),n4w8x=$(s 4 2 $X),n4w16=$(s 4 4 0),n4w16x=$(s 4 4 $X),n2w8x=$(s 2 2 $X),tiled=2,t32=$((2|(2<<4))),xk2=$((4|(2<<4))),auto2=0"
sh=""
for kn in ${KN:-4096x4096 4096x12288 4096x22016 11008x4096 4096x6144 4096x28672 14336x4096 8192x8192 8192x10240 28672x8192 5120x5120 5120x13824}; do for m in ${MS:-17 20 24 32 40 48 56 64}; do sh="$sh,${m}x$kn"; done; done
timeout 2400 python tools/wide_probe.py --shapes "${sh:1}" --variants "$V" --iters 24 --out gpurun_out/mid_audit.jsonl 2>&1 | grep -v amdgpu.ids | tail -1
