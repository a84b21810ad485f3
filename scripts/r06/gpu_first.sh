#!/bin/bash
# first evidence run of the r06 tree: XM parity tests, XM against the other families (dispatch clock + span), phase stamps, bench line
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gemm_gpu.py -q -m gpu -k "xm" -x > gpurun_out/r06/pytest_xm.txt 2>&1; tail -3 gpurun_out/r06/pytest_xm.txt
timeout 600 python tools/xm_check.py --no-check 64x4096x4096 64x4096x12288 64x4096x22016 64x11008x4096 48x4096x22016 32x4096x4096 24x4096x4096 64x4096x6144 64x8192x8192 2>&1 | grep "   " | cut -c1-170 > gpurun_out/r06/xm_time_first.txt
cat gpurun_out/r06/xm_time_first.txt
QUICK_AMD_LIB_OVERRIDE=$PWD/tools/bin/libquick_amd_tools.so timeout 600 python tools/xm_phases.py --pr 1,3 64x4096x4096 64x4096x22016 > gpurun_out/r06/xm_phases_first.txt 2>&1
QUICK_AMD_LIB_OVERRIDE=$PWD/tools/bin/libquick_amd_tools.so timeout 600 python tools/xm_phases.py --t32 --pr 1,2 64x4096x4096 >> gpurun_out/r06/xm_phases_first.txt 2>&1
cat gpurun_out/r06/xm_phases_first.txt
timeout 900 python bench.py > gpurun_out/r06/bench_first.json 2> gpurun_out/r06/bench_first.err; tail -c 3000 gpurun_out/r06/bench_first.json
