#!/bin/bash
# the planner's new pick at 129..256 tokens on N = 4096 layers (128 x 128 tile, four slices) against the old one (64 x 128, two slices), one session; then the GPU suite
mkdir -p gpurun_out/r06
S="160x4096x4096,192x4096x4096,256x4096x4096,144x4096x4096,160x11008x4096,192x11008x4096,224x14336x4096,160x8192x4096"
timeout 900 python tools/wide_probe.py --shapes $S --variants auto=0,xw21s2=0x225,xw41s4=0x1445,auto2=0 --iters 20 --out gpurun_out/r06/rule414.jsonl > gpurun_out/r06/rule414.log 2>&1
grep "auto\|xw21s2\|xw41s4" gpurun_out/r06/rule414.log | cut -c1-150
(timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -3) | tee gpurun_out/r06/rule414_pytest.txt
