#!/bin/bash
# lean kernels on / off in one session: per-launch times of a decode layer's launches (64-deep hipGraph chains) and decode tok/s
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05
out=gpurun_out/r05/ab_decode.txt; : > $out
for r in 1 2; do for lean in 0 1; do
  echo "== QUICK_AMD_LEAN=$lean (round $r)" >> $out
  for b in 1 4 16; do QUICK_AMD_LEAN=$lean timeout 300 python tools/time_ops.py $b 2>/dev/null | grep -v "^B=.*\(rmsnorm  \|silu_mul  \)" >> $out; done
  QUICK_AMD_LEAN=$lean timeout 600 python bench_decode.py --model llama2-7b --bs 1 4 16 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    print('decode', d['model'], 'bs', d['batch'], round(d['decode_tok_s'], 1), 'tok/s', round(d['decode_ms_per_step'], 4), 'ms')" >> $out
done; done
cat $out
