#!/bin/bash
# usage (GPU box): tools/ab.sh "<names>" [rounds]   -- alternates the variant libraries built by tools/build_variant.sh
# over the decode-layer GEMM shapes and the bs=1/8 decode step, so that box-to-box and drift noise cancels
cd "$(dirname "$0")/.."
names=$1; rounds=${2:-2}
for r in $(seq $rounds); do
for n in $names; do
  export QUICK_AMD_LIB_OVERRIDE=$PWD/quick_amd/lib/ab_$n.so
  echo "== $n (round $r)"
  for s in "1 4096 22016" "1 4096 12288" "1 11008 4096" "1 4096 4096" "8 4096 22016" "8 4096 4096" "16 4096 12288"; do
    set -- $s
    python bench.py --steps 40 --warmup 5 --cpu-seconds 0 --K $2 --N $3 --sweep $1 --M $1 --sets 8 2>&1 >/dev/null | grep "M=" | sed 's/roofline.*//; s/(cache.*//' | sed "s/^/K=$2 N=$3 /"
  done
  python bench_decode.py --model llama2-7b --bs 1 8 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    print('decode bs', d['batch'], round(d['decode_tok_s'], 1), 'tok/s', round(d['decode_ms_per_step'], 4), 'ms')"
done; done
