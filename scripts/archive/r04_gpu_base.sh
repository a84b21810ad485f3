#!/bin/bash
# r04 baseline: phase stamps of the r02 128 x 256 four-wave tile with 4 K slices (the loop the new kernel must beat) and of the current pick
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04
export QUICK_AMD_LIB_OVERRIDE=$PWD/quick_amd/lib/libquick_amd_tools.so
{
python tools/wide_phases.py --kernel $((3 + (4<<4) + (2<<8) + (1<<12))) 512x4096x4096 512x4096x4096
python tools/wide_phases.py --kernel $((3 + (4<<4) + (2<<8))) 512x4096x4096
python tools/xk_phases.py --kernel $((4 + (2<<4) + (1<<8))) 512x4096x4096
python tools/xk_phases.py --kernel $((4 + (4<<4) + (2<<8))) 512x4096x4096
} > gpurun_out/r04/base_phases.txt 2>&1
tail -60 gpurun_out/r04/base_phases.txt
