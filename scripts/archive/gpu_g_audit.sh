#!/bin/bash
# planner audit at group sizes 64 and 32 (BASELINE uses 128): families and skinny flavours on a few model shapes
cd "$GRAFT_REPO_ROOT" || exit 1
for G in 64 32; do
python tools/wide_probe.py --G $G --shapes ${1:-1x4096x4096,8x4096x4096,16x4096x12288,64x4096x4096,64x4096x12288,512x4096x4096,2048x4096x4096,1x11008x4096,16x11008x4096,512x11008x4096} --variants "auto=0,skinny=1,exact=$((1+(1<<25))),dz=$((1+(1<<26))),ntw4=$((1+(4<<4))),tiled=2,tiled32=$((2+(2<<4))),tiledwide=$((2+(1<<29)))" --iters 30 --out gpurun_out/g_audit_$G.jsonl 2>&1 | grep -v amdgpu.ids | tail -1
done
