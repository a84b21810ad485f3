#!/bin/bash
# one iteration on the mid-token loop: correctness, phase stamps (tools library), timing against the other families
mkdir -p gpurun_out/r06
tag=${1:-iter}
timeout 600 python tools/xm_check.py --no-time 64x4096x4096 33x4096x4096 17x1024x256 50x1536x4096 40x11008x512 > gpurun_out/r06/xm_check_$tag.txt 2>&1
echo "wrong: $(grep -c WRONG gpurun_out/r06/xm_check_$tag.txt)"; tail -1 gpurun_out/r06/xm_check_$tag.txt
QUICK_AMD_LIB_OVERRIDE=$PWD/tools/bin/libquick_amd_tools.so timeout 600 python tools/xm_phases.py ${XM_PHASE_ARGS:---pr 1,3 64x4096x4096 64x4096x22016} > gpurun_out/r06/xm_phases_$tag.txt 2>&1
cat gpurun_out/r06/xm_phases_$tag.txt
timeout 600 python tools/xm_check.py --no-check ${XM_SHAPES:-64x4096x4096 64x4096x12288 64x4096x22016 64x11008x4096 24x4096x4096} 2>&1 | grep "   " | grep -v big | cut -c1-140 > gpurun_out/r06/xm_time_$tag.txt
cat gpurun_out/r06/xm_time_$tag.txt
