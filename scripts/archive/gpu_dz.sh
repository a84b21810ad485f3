#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q -k "skinny or deferred or synthetic_sweep or rmsnorm or model_shapes or baseline" 2>&1 | tail -3
sh=""
for kn in 4096x4096 4096x12288 4096x22016 11008x4096 4096x6144 8192x8192; do for m in 2 3 4 6 8 12 16; do sh="$sh,${m}x$kn"; done; done
timeout 1500 python tools/wide_probe.py --shapes "${sh:1}" --variants "warm=0,auto=0,dz=$((1+(1<<26))),exact=$((1+(1<<25)))" --iters 40 --out gpurun_out/dz_probe.jsonl 2>&1 | grep -v amdgpu.ids | cut -c1-150 | tee gpurun_out/dz_probe.txt | tail -5
