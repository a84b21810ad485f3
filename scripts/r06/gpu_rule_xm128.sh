#!/bin/bash
# the mid-token rule at 96..128 tokens with K <= 8192 (two 64-token tiles x 2 pairs) against the launches it replaces, one session; then the GPU suite
mkdir -p gpurun_out/r06
S="96x8192x8192,112x8192x8192,128x8192x8192,128x8192x7168,128x5120x8192,112x6144x8192"
timeout 900 python tools/wide_probe.py --shapes $S --variants auto=0,xw41s4=0x1445,xw21s2=0x225,xw41s2=0x1245,auto2=0 --iters 20 --out gpurun_out/r06/rule_xm128.jsonl > gpurun_out/r06/rule_xm128.log 2>&1
grep "auto\|xw21s2\|xw41s" gpurun_out/r06/rule_xm128.log | cut -c1-150
(timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -3) | tee gpurun_out/r06/rule_xm128_pytest.txt
