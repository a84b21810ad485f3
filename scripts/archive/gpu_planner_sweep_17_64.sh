#!/bin/bash
# planner audit, 17..64 tokens (two to four 16-token blocks): tuned and untuned layer shapes x skinny tile counts, tiled, wide
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
KN="4096x4096 4096x8192 4096x12288 4096x22016 11008x4096 4096x6144 4096x28672 14336x4096 8192x8192 8192x10240 8192x57344 28672x8192 5120x5120 5120x15360 5120x27648 13824x5120 3584x4608 3584x37888 18944x3584 7168x7168 7168x40960 20480x7168"
sh=""
for kn in $KN; do for m in 20 24 32 40 48 56 64; do sh="$sh,${m}x$kn"; done; done
N4=$((1+(4<<4))); N2=$((1+(2<<4)))
python tools/wide_probe.py --shapes "${sh:1}" --variants "warm=0,auto=0,ntw2=$N2,ntw4=$N4,tiled=2,tiled32=$((2+(2<<4))),w2x1=$((3+32+256)),w2x1e=$((3+32+256+(1<<15)+(4<<22)))" --iters 16 --out gpurun_out/planner_sweep_17_64.jsonl 2>&1 | grep -v amdgpu.ids | tail -1
