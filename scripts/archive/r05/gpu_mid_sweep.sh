#!/bin/bash
# 33..128 tokens on the model layers: the planner's pick against forced xw / xk / skinny / tiled launches (dispatch clock, HBM-cold sets)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05
shapes=""
for kn in 4096x4096 4096x6144 4096x12288 4096x22016 4096x28672 11008x4096 14336x4096 8192x8192 8192x10240 28672x8192 5120x5120 5120x13824 13824x5120; do for m in 33 48 64 96 128; do shapes="$shapes,${m}x$kn"; done; done
shapes=${shapes#,}
timeout 2400 python tools/wide_probe.py --iters 30 --shapes $shapes --variants auto=0,xw2s1=0x125,xw2s2=0x225,xw41s1=0x1145,xw41s2=0x1245,xw41s4=0x1445,xk2=0x24,xk2s1=0x124,xk2s2=0x224,xk2s4=0x424,xk4=0x44,skinny=1,tiled=2 --out gpurun_out/r05/mid_sweep.jsonl > gpurun_out/r05/mid_sweep.txt 2>&1
tail -5 gpurun_out/r05/mid_sweep.txt
