#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
W21=$((3+32+256)); W82=$((3+128+512)); NR=4096; A1=$((1<<16)); A2=$((2<<16))
V="w2x1nr=$((W21+NR)),w2x1nr_loadonly=$((W21+NR+A1)),w2x1nr_computeonly=$((W21+NR+A2)),w2x1r6=$((W21)),w2x1r6_loadonly=$((W21+A1)),w2x1r6_computeonly=$((W21+A2))"
python tools/wide_probe.py --shapes 512x4096x4096,2048x4096x4096 --variants "$V" --iters 40 2>&1 | grep -v amdgpu.ids | cut -c1-150
V="w8x2=$((W82)),w8x2_loadonly=$((W82+A1)),w8x2_computeonly=$((W82+A2))"
python tools/wide_probe.py --shapes 4096x8192x8192 --variants "$V" --iters 20 2>&1 | grep -v amdgpu.ids | cut -c1-150
