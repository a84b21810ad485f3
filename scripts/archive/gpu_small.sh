#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
W21=$((3+32+256)); NR=4096
V="auto=0,w2x1r:8=$((W21)):8,w2x1r:4=$((W21)):4,w2x1nr:8=$((W21+NR)):8,w2x1nr:4=$((W21+NR)):4,tiled=2,tiled:8=2:8,skinny=1"
python tools/wide_probe.py --shapes 64x4096x4096,32x4096x4096,48x4096x4096,64x4096x11008,64x11008x4096 --variants "$V" --iters 40 2>&1 | grep -v amdgpu.ids | cut -c1-150
V="auto=0,skinny:2=1:2,skinny:3=1:3,skinny:4=1:4"
python tools/wide_probe.py --shapes 1x11008x4096,1x14336x4096,1x28672x8192,8x11008x4096,16x11008x4096 --variants "$V" --iters 60 2>&1 | grep -v amdgpu.ids | cut -c1-150
