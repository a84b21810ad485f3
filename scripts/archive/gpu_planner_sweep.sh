#!/bin/bash
# planner audit: every model shape x a ladder of token counts x the kernel families (and the skinny flavours at small M);
# writes gpurun_out/planner_sweep_{small,big}.jsonl (kept as profiles/r02_planner_audit*.jsonl)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
if [ "$1" = "tests" ]; then timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -3; fi
KN="4096x4096 4096x12288 4096x22016 11008x4096 4096x6144 4096x28672 14336x4096 8192x8192 8192x10240 8192x57344 28672x8192"
small=""; big=""
for kn in $KN; do for m in 2 3 4 6 8 12 16; do small="$small,${m}x$kn"; done; for m in 24 32 48 64 96 128 192 256 384; do big="$big,${m}x$kn"; done; done
EX=$((1+(1<<25))); DZ=$((1+(1<<26))); N4=$((1+(4<<4))); N1=$((1+(1<<4))); N2=$((1+(2<<4)))
python tools/wide_probe.py --shapes "${small:1}" --variants "auto=0,exact=$EX,dz=$DZ,ntw1=$N1,ntw2=$N2,ntw4=$N4,tiled=2" --iters 24 --out gpurun_out/planner_sweep_small.jsonl 2>&1 | grep -v amdgpu.ids | tail -1
python tools/wide_probe.py --shapes "${big:1}" --variants "auto=0,skinny=1,tiled=2,tiled32=$((2+(2<<4))),wide=3,w2x1=$((3+32+256)),w4x1=$((3+64+256))" --iters 24 --out gpurun_out/planner_sweep_big.jsonl 2>&1 | grep -v amdgpu.ids | tail -1
