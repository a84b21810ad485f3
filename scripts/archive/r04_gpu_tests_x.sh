#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04
timeout 1200 python -m pytest tests -q -m gpu -k "xk or xw or pin or golden_fixture_forward" -x > gpurun_out/r04/pytest_x.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r04/pytest_x.log
tail -30 gpurun_out/r04/pytest_x.log
timeout 300 python tools/wide_probe.py --shapes 128x4096x4096,64x11008x4096,64x4096x12288,256x4096x4096 --variants auto=0,xk64=0x24,xk128=0x44 --iters 60
