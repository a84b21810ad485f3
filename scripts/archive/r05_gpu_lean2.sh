#!/bin/bash
# lean small-M kernels, second look: parity + spans against the planner's pick, phase anatomy
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05
timeout 900 python tools/lean_check.py 1x4096x4096 2x4096x4096 4x4096x4096 8x4096x4096 16x4096x4096 1x4096x12288 4x4096x12288 1x4096x22016 4x4096x22016 8x4096x22016 1x11008x4096 4x11008x4096 8x11008x4096 1x8192x8192 > gpurun_out/r05/lean_check2.txt 2>&1
QUICK_AMD_LIB_OVERRIDE=tools/bin/libquick_amd_tools.so timeout 600 python tools/lean_phases.py 1x4096x4096 8x4096x4096 16x4096x4096 1x4096x22016 1x11008x4096 > gpurun_out/r05/lean_phases2.txt 2>&1
cat gpurun_out/r05/lean_check2.txt
cat gpurun_out/r05/lean_phases2.txt
