#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
W21=$((3+32+256)); W41=$((3+64+256)); W22=$((3+32+512)); W42=$((3+64+512)); NR=4096
V="tiled=2,tiled:2=2:2,w2x1nr=$((W21+NR)),w2x1nr:2=$((W21+NR)):2,w2x1nr:4=$((W21+NR)):4,w2x1n3:2=$((W21+(3<<22))):2,w2x1n4:2=$((W21+(4<<22))):2,w2x1n6:2=$((W21)):2,w4x1nr:2=$((W41+NR)):2,w4x1nr:4=$((W41+NR)):4,w4x1n3:4=$((W41)):4,w2x2nr:2=$((W22+NR)):2,w2x2nr:4=$((W22+NR)):4,w4x2nr:4=$((W42+NR)):4,w4x2nr:8=$((W42+NR)):8"
python tools/wide_probe.py --shapes 512x4096x4096,256x4096x4096,1024x4096x4096 --variants "$V" --iters 40 2>&1 | grep -v amdgpu.ids | cut -c1-150
