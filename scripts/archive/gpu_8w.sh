#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 600 python -m pytest tests/test_gemm_gpu.py -x -q -m gpu -k "synthetic_sweep or golden_fixture_forward or prefill_shapes" 2>&1 | tail -3
W21=$((3+32+256)); W41=$((3+64+256)); W22=$((3+32+512)); NR=4096; E=$((1<<15)); A1=$((1<<16)); A2=$((2<<16))
V="tiled=2,w2x1nr=$((W21+NR)),w2x1r6=$((W21)),w2x1e6=$((W21+E)),w2x1e4=$((W21+E+(4<<22))),w2x1e3=$((W21+E+(3<<22))),w2x1e6_loadonly=$((W21+E+A1)),w2x1e6_computeonly=$((W21+E+A2)),w2x1e6:2=$((W21+E)):2,w2x2e=$((W22+E)),w4x1e=$((W41+E)),w2x2e:2=$((W22+E)):2,w4x1e:2=$((W41+E)):2"
python tools/wide_probe.py --shapes 512x4096x4096,1024x4096x4096,2048x4096x4096 --variants "$V" --iters 40 2>&1 | grep -v amdgpu.ids | cut -c1-150
