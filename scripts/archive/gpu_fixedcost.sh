#!/bin/bash
# where the K-independent part of a large-tile launch goes: phase stamps of the 256 x 256 tile + the prefill shapes
cd "$GRAFT_REPO_ROOT" || exit 1
python tools/wide_phases.py 256x128x4096 4096x128x4096 4096x4096x4096 2>&1 | grep -v amdgpu.ids
python -m pytest tests/test_gemm_gpu.py -x -q -m gpu -k "wide or epilogue or fus or random" 2>&1 | tail -5
python tools/wide_probe.py --shapes 512x4096x4096,1024x4096x4096,2048x4096x4096,4096x4096x4096,4096x4096x11008,4096x11008x4096,8192x8192x8192 --variants "auto=0" --iters 30 2>&1 | grep -v amdgpu.ids | cut -c1-120
