#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
out=gpurun_out/r04c; mkdir -p $out
timeout 2400 python -m pytest tests -m gpu -x -q > $out/gpu_tests.txt 2>&1; tail -3 $out/gpu_tests.txt
python bench.py --decode-seconds 0 --cpu-seconds 0 > $out/bench_nodecode.json 2>$out/bench_err.log; python - <<'P'
import json
d=json.loads(open('gpurun_out/r04c/bench_nodecode.json').read().strip().splitlines()[-1])
print('M=512', d['ms_per_step']*1e3, d['roofline']['kernel_us'], d['roofline']['frac'])
for p in d['prefill_layers']: print(p['M'],p['K'],p['N'], round(p['kernel_us'],1), round(p['roofline']['frac'],3), p['plan'][:40])
P
