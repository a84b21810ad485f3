#!/bin/bash
# audit of the planner after the exchange-K kernels joined its cost model: its pick against forced families -> gpurun_out/xk_audit.jsonl
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
KN=${KN:-"4096x4096 4096x12288 4096x22016 11008x4096 4096x6144 4096x28672 14336x4096 8192x8192 8192x10240 28672x8192 5120x5120 5120x13824 13824x5120 4096x14336 7168x7168 3584x18944"}
MS=${MS:-"40 48 64 80 96 128 160 192 256 320 384 512 640 768 1024 2048"}
sh=""
for kn in $KN; do for m in $MS; do sh="$sh,${m}x$kn"; done; done
XK=4
v() { echo $(( XK | ($1 << 4) | ($2 << 8) )); }
timeout 3000 python tools/wide_probe.py --shapes "${sh:1}" --variants "warm=0,auto=0,tiled=2,wide=3,w2x1e=$((3+32+256+(1<<15)+(4<<22))),w4x1=$((3+64+256)),w4x2=$((3+64+512)),xk2=$(v 2 0),xk4=$(v 4 0)" --iters ${ITERS:-16} --out gpurun_out/xk_audit.jsonl 2>&1 | grep -v amdgpu.ids | tail -2
