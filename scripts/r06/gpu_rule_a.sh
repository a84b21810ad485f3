#!/bin/bash
# 57..64 tokens on 8192 x 10240 (long K, 160 workgroups of 64 tokens x 64 channels): the planner's new pick (64 x 128 four-wave tile, two slices) against the mid-token launch; then the GPU suite
mkdir -p gpurun_out/r06
timeout 600 python tools/wide_probe.py --shapes 57x8192x10240,60x8192x10240,64x8192x10240,64x6144x10240,56x8192x10240 --variants auto=0,xm22=0x227,xw21s2=0x225,auto2=0 --iters 20 --out gpurun_out/r06/rule_a.jsonl > gpurun_out/r06/rule_a.log 2>&1
grep "auto\|xm22\|xw21s2" gpurun_out/r06/rule_a.log | cut -c1-150
(timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -3) | tee gpurun_out/r06/rule_a_pytest.txt
