#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05
timeout 600 python -m pytest tests -q -m gpu -x -k "xw or mid_token" 2>&1 | tail -3
timeout 600 python bench.py --cpu-seconds 0 --decode-seconds 0 --layers "" --steps 20 --warmup 5 2>&1 >/dev/null | grep "M=" 
timeout 600 python bench.py --cpu-seconds 0 --decode-seconds 0 --layers "" 2>&1 >/dev/null | grep "M="
shapes=""
for kn in 4096x4096 4096x6144 5120x5120 8192x8192 13824x5120 28672x8192; do for m in 96 128; do shapes="$shapes,${m}x$kn"; done; done
timeout 600 python tools/wide_probe.py --iters 30 --shapes ${shapes#,} --variants auto=0 2>&1 | grep auto | cut -c1-150
export QUICK_AMD_LIB_OVERRIDE=$PWD/tools/bin/libquick_amd_tools.so
timeout 120 python tools/xk_phases.py --kernel 0x1205 512x4096x4096 2>&1 | grep -v amdgpu | head -12
