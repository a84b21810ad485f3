#!/bin/bash
# 5..16 tokens on the model layers: the planner's pick against forced skinny flavours (a warm-up variant first)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
SK=1
s() { echo $(( SK | ($1 << 4) | ($2 << 8) | $3 )); }
X=$((1<<12))
V="warm=0,auto=0,n1w8=$(s 1 2 0),n1w8x=$(s 1 2 $X),n1w16=$(s 1 4 0),n1w16x=$(s 1 4 $X),n2w8x=$(s 2 2 $X),n2w16x=$(s 2 4 $X),n4w8x=$(s 4 2 $X),n4w16x=$(s 4 4 $X),tab=$(s 1 2 $((1<<26))),tab16=$(s 1 4 $((1<<26))),auto2=0"
sh=""
for kn in ${KN:-4096x4096 4096x12288 4096x22016 11008x4096 4096x6144 4096x28672 14336x4096 8192x8192 8192x10240 28672x8192 5120x5120 8192x57344}; do for m in ${MS:-5 6 8 10 12 16}; do sh="$sh,${m}x$kn"; done; done
timeout 2000 python tools/wide_probe.py --shapes "${sh:1}" --variants "$V" --iters 24 --out gpurun_out/small_audit2.jsonl 2>&1 | grep -v amdgpu.ids | tail -1
