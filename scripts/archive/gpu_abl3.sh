#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 600 python -m pytest tests/test_gemm_gpu.py -x -q -m gpu -k "synthetic_sweep or golden_fixture_forward or prefill_shapes" 2>&1 | tail -3
W21=$((3+32+256)); W82=$((3+128+512)); W42=$((3+64+512)); NR=4096; A1=$((1<<16)); A2=$((2<<16))
V="tiled=2,w2x1nr=$((W21+NR)),w2x1nr_computeonly=$((W21+NR+A2)),w2x1r6=$((W21)),w2x1r4=$((W21+(4<<22))),w2x1r6_loadonly=$((W21+A1)),w2x1r6_computeonly=$((W21+A2))"
python tools/wide_probe.py --shapes 512x4096x4096,1024x4096x4096 --variants "$V" --iters 40 2>&1 | grep -v amdgpu.ids | cut -c1-150
V="tiled=2,w4x2nr=$((W42+NR)),w4x2r3=$((W42)),w8x2=$((W82)),w8x2_computeonly=$((W82+A2))"
python tools/wide_probe.py --shapes 2048x4096x4096,4096x8192x8192 --variants "$V" --iters 20 2>&1 | grep -v amdgpu.ids | cut -c1-150
