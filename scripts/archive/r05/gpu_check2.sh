#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05
timeout 2700 python -m pytest tests -q -m gpu -x 2>&1 | tail -3
timeout 600 python bench_decode.py --model llama2-7b mistral-7b --bs 64 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    print('decode', d['model'], 'bs', d['batch'], round(d['decode_tok_s'], 1), 'tok/s', round(d['decode_ms_per_step'], 4), 'ms')"
timeout 300 python tools/time_ops.py 64 2>/dev/null | grep gemm
