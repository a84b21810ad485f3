#!/bin/bash
# planner audit, 64..8192 tokens: model shapes x every wide kernel the planner can pick (ring / double-buffered) and the r01 tiled kernel
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
KN="4096x4096 4096x12288 4096x22016 11008x4096 4096x6144 4096x28672 14336x4096 8192x8192 8192x10240 8192x57344 28672x8192"
MS=${MS:-"64 96 128 192 256 320 384 448 512 640 768 1024 1280 1536 2048 3072 4096 8192"}
sh=""
for kn in $KN; do for m in $MS; do sh="$sh,${m}x$kn"; done; done
NR=4096
W21=$((3+32+256)); W22=$((3+32+512)); W41=$((3+64+256)); W42=$((3+64+512)); W81=$((3+128+256)); W82=$((3+128+512))
python tools/wide_probe.py --shapes "${sh:1}" --variants "warm=0,auto=0,tiled=2,w2x1=$W21,w2x1e=$((W21+(1<<15)+(4<<22))),w2x1nr=$((W21+NR)),w2x2=$W22,w2x2nr=$((W22+NR)),w4x1=$W41,w4x1nr=$((W41+NR)),w4x2=$W42,w4x2nr=$((W42+NR)),w8x1=$W81,w8x2=$W82" --iters ${ITERS:-12} --out gpurun_out/planner_sweep_large.jsonl 2>&1 | grep -v amdgpu.ids | tail -1
