#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests -q -m gpu -x -k "lean" 2>&1 | tail -4
timeout 1500 python tools/lean_check.py --no-check --persist 1x4096x12288 4x4096x12288 8x4096x12288 16x4096x12288 1x4096x22016 4x4096x22016 8x4096x22016 16x4096x22016 1x4096x28672 1x4096x32000 1x8192x57344 > gpurun_out/r05/lean_persist1.txt 2>&1
cat gpurun_out/r05/lean_persist1.txt
