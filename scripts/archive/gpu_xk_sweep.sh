#!/bin/bash
# exchange-K kernels against the planner's choice, 24..1024 tokens on the model layer shapes -> gpurun_out/xk_sweep.jsonl
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
KN=${KN:-"4096x4096 4096x12288 4096x22016 11008x4096 4096x6144 4096x28672 14336x4096 8192x8192 8192x10240 8192x57344 28672x8192 5120x5120 5120x13824 13824x5120 4096x14336"}
MS=${MS:-"24 32 48 64 96 128 160 192 256 320 384 448 512 640 768 1024"}
sh=""
for kn in $KN; do for m in $MS; do sh="$sh,${m}x$kn"; done; done
XK=4
v() { echo $(( XK | ($1 << 4) | ($2 << 8) )); }
timeout 3000 python tools/wide_probe.py --shapes "${sh:1}" --variants "warm=0,auto=0,xk2=$(v 2 0),xk4=$(v 4 0),xk2h=$(v 2 15),xk4h=$(v 4 15)" --iters ${ITERS:-16} --out gpurun_out/${OUT:-xk_sweep}.jsonl 2>&1 | grep -v amdgpu.ids | tail -2
