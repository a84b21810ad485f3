#!/bin/bash
# 1..4 tokens on the model layers: the planner's pick against forced skinny flavours (table / exact, 8 / 16 waves, persistence)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
SK=1
s() { echo $(( SK | ($1 << 4) | ($2 << 8) | $3 )); }
X=$((1<<12)); T=$((1<<26))
V="warm=0,auto=0,tab=$(s 1 2 $T),tab16=$(s 1 4 $T),tab4=$(s 1 1 $T),tabs1=$(s 1 2 $((T|(1<<22)))),tabs2=$(s 1 2 $((T|(2<<22)))),tabflip=$(s 1 2 $((T|(1<<21)))),n1w8x=$(s 1 2 $X),n1w16x=$(s 1 4 $X),n2w8x=$(s 2 2 $X),n4w8x=$(s 4 2 $X),auto2=0"
sh=""
for kn in ${KN:-4096x4096 4096x12288 4096x22016 11008x4096 4096x6144 4096x28672 14336x4096 8192x8192 8192x10240 28672x8192 5120x5120 8192x57344}; do for m in ${MS:-1 2 3 4}; do sh="$sh,${m}x$kn"; done; done
timeout 2400 python tools/wide_probe.py --shapes "${sh:1}" --variants "$V" --iters 24 --out gpurun_out/tiny_audit.jsonl 2>&1 | grep -v amdgpu.ids | tail -1
