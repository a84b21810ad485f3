#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for r in 1 2; do for n in span nospan; do
  export QUICK_AMD_LIB_OVERRIDE=$PWD/quick_amd/lib/ab_$n.so
  echo "== $n (round $r)"
  python bench.py --steps 400 --warmup 20 --cpu-seconds 0 --decode-seconds 0 --sweep 1,8 --M 1 2>&1 >/dev/null | grep -E "^M=|layer" | cut -c1-100
  python bench_decode.py --model llama2-7b --bs 1 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('decode bs', d['batch'], round(d['decode_tok_s'], 1), 'tok/s')"
done; done
