#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05
export QUICK_AMD_LIB_OVERRIDE=tools/bin/libquick_amd_tools.so
timeout 600 python tools/lean_phases.py --waves 8 1x4096x12288 > gpurun_out/r05/lean_phases6.txt 2>&1
timeout 600 python tools/lean_phases.py --ln --waves 8 1x4096x12288 >> gpurun_out/r05/lean_phases6.txt 2>&1
cat gpurun_out/r05/lean_phases6.txt
