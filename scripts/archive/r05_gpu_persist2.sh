#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -4
timeout 900 python tools/lean_check.py --no-check --planner-only 8x4096x12288 16x4096x12288 8x4096x22016 12x4096x22016 16x4096x22016 16x4096x28672 > gpurun_out/r05/lean_persist2.txt 2>&1; cat gpurun_out/r05/lean_persist2.txt
for lean in 0 1; do echo "== QUICK_AMD_LEAN=$lean"; QUICK_AMD_LEAN=$lean timeout 600 python bench_decode.py --model llama2-7b mistral-7b --bs 8 16 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    print('decode', d['model'], 'bs', d['batch'], round(d['decode_tok_s'], 1), 'tok/s', round(d['decode_ms_per_step'], 4), 'ms')"; done
