#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 600 python -m pytest tests/test_gemm_gpu.py -x -q -k "chain" 2>&1 | tail -15
for m in llama2-7b; do
  timeout 300 python bench_decode.py --model $m --bs 1 2 4 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.rstrip()); continue
    print(d['model'], d['batch'], 'chain' if d['chained_gemms'] else 'single', round(d['decode_tok_s'], 1), 'tok/s', round(d['decode_ms_per_step'], 3), 'ms')"
  timeout 300 python bench_decode.py --model $m --bs 1 2 4 --no-chain 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.rstrip()); continue
    print(d['model'], d['batch'], 'chain' if d['chained_gemms'] else 'single', round(d['decode_tok_s'], 1), 'tok/s', round(d['decode_ms_per_step'], 3), 'ms')"
done
