#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests -q -m gpu -x -k "lean or rmsnorm or fused" > gpurun_out/r05/pytest_lean.log 2>&1; tail -5 gpurun_out/r05/pytest_lean.log
bash scripts/r05/gpu_ab_fused.sh | tail -16
timeout 1200 python tools/lean_check.py 1x4096x4096 4x4096x4096 1x4096x12288 4x4096x12288 1x4096x22016 4x4096x22016 1x11008x4096 4x11008x4096 1x8192x8192 1x4096x1024 1x8192x1024 > gpurun_out/r05/lean_check7.txt 2>&1
cat gpurun_out/r05/lean_check7.txt
