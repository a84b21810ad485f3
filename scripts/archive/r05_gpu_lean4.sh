#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05
QUICK_AMD_LIB_OVERRIDE=tools/bin/libquick_amd_tools.so timeout 600 python tools/lean_phases.py --waves 8,16 1x4096x4096 > gpurun_out/r05/lean_phases4.txt 2>&1
cat gpurun_out/r05/lean_phases4.txt
