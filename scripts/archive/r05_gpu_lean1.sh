#!/bin/bash
# first look at the lean small-M kernels: ramp probe, parity + spans against the planner's pick, phase anatomy
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05
timeout 120 tools/bin/ramp_probe > gpurun_out/r05/ramp_probe.txt 2>&1
timeout 600 python tools/lean_check.py > gpurun_out/r05/lean_check1.txt 2>&1
QUICK_AMD_LIB_OVERRIDE=tools/bin/libquick_amd_tools.so timeout 600 python tools/lean_phases.py > gpurun_out/r05/lean_phases1.txt 2>&1
tail -60 gpurun_out/r05/lean_check1.txt
tail -70 gpurun_out/r05/lean_phases1.txt
tail -30 gpurun_out/r05/ramp_probe.txt
