#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05
python tools/dbg_persist.py 2>&1 | grep -v amdgpu | grep -c "n bad 0"
python tools/dbg_persist.py 2>&1 | grep -v "n bad 0" | grep -v amdgpu | head
timeout 900 python -m pytest tests -q -m gpu -x -k "lean or fragment_deferred" 2>&1 | tail -3
timeout 1500 python tools/lean_check.py --no-check --persist 8x4096x12288 16x4096x12288 8x4096x22016 12x4096x22016 16x4096x22016 1x4096x22016 4x4096x22016 > gpurun_out/r05/lean_persist3.txt 2>&1
cat gpurun_out/r05/lean_persist3.txt
