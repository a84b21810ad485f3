#!/bin/bash
# planner audit, 24..448 tokens: model shapes x the families and the wide kernels the planner can pick -> gpurun_out/planner_sweep_mid.jsonl
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
KN="4096x4096 4096x12288 4096x22016 11008x4096 4096x6144 4096x28672 14336x4096 8192x8192 8192x10240 8192x57344 28672x8192 5120x5120 5120x13824 13824x5120 4096x14336"
MS=${MS:-"24 32 48 64 80 96 112 128 160 192 224 256 320 384 448"}
sh=""
for kn in $KN; do for m in $MS; do sh="$sh,${m}x$kn"; done; done
NR=4096
W21=$((3+32+256)); W22=$((3+32+512)); W41=$((3+64+256)); W42=$((3+64+512)); W82=$((3+128+512))
# "warm" = the planner's choice once more (the first variant of a shape reads high: clocks ramp after the allocation pause)
python tools/wide_probe.py --shapes "${sh:1}" --variants "warm=0,auto=0,skinny=1,tiled=2,tiled32=$((2+(2<<4))),w2x1=$W21,w2x1e=$((W21+(1<<15)+(4<<22))),w2x1nr=$((W21+NR)),w2x2nr=$((W22+NR)),w4x1nr=$((W41+NR)),w4x2nr=$((W42+NR)),w8x2=$W82" --iters ${ITERS:-16} --out gpurun_out/planner_sweep_mid.jsonl 2>&1 | grep -v amdgpu.ids | tail -1
