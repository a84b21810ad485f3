#!/bin/bash
# small-M audit on the final tree: the planner's pick against forced skinny flavours at 6..64 tokens
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
SK=1
s() { echo $(( SK | ($1 << 4) | ($2 << 8) | $3 )); }   # channel tiles per wave, waves / 4, extra bits
V="auto=0"
for nt in 1 2 4; do for w in 2 4; do
  V="$V,n${nt}w$((w*4))=$(s $nt $w 0),n${nt}w$((w*4))x=$(s $nt $w $((1<<12)))"
done; done
V="$V,tab=$(s 1 2 $((1<<26))),tab16=$(s 1 4 $((1<<26))),exact=$(s 0 0 $((1<<25))),xk2=$((4|(2<<4))),tiled=2"
sh=""
for kn in 4096x4096 4096x12288 11008x4096 4096x22016; do for m in 6 8 12 16 24 32 48 64; do sh="$sh,${m}x$kn"; done; done
timeout 1500 python tools/wide_probe.py --shapes "${sh:1}" --variants "$V" --iters 24 --out gpurun_out/small_audit.jsonl 2>&1 | grep -v amdgpu.ids | tail -2
