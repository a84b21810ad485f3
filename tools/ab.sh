#!/bin/bash
# usage (GPU box): tools/ab.sh "<names>" [rounds]   -- alternates the variant libraries built by tools/build_variant.sh
# over the decode-layer GEMM shapes and the bs=1/8 decode step, so that box-to-box and drift noise cancels
cd "$(dirname "$0")/.."
names=$1; rounds=${2:-2}
# AB_SHAPES: newline-separated "M K N" triples; AB_DECODE: bench_decode.py arguments
DEFAULT_SHAPES=$'1 4096 22016\n1 4096 12288\n1 11008 4096\n1 4096 4096\n8 4096 22016\n8 4096 4096\n16 4096 12288'
for r in $(seq $rounds); do
for n in $names; do
  export QUICK_AMD_LIB_OVERRIDE=$PWD/tools/bin/ab_$n.so
  echo "== $n (round $r)"
  while read -r s; do
    [ -z "$s" ] && continue
    set -- $s
    python bench.py --steps 40 --warmup 5 --cpu-seconds 0 --layers "" --K $2 --N $3 --sweep $1 --M $1 --sets 8 2>&1 >/dev/null | grep "M=" | sed 's/roofline.*//; s/(cache.*//' | sed "s/^/K=$2 N=$3 /"
  done <<< "${AB_SHAPES:-$DEFAULT_SHAPES}"
  python bench_decode.py ${AB_DECODE:---model llama2-7b --bs 1 8} 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    print('decode', d['model'], 'bs', d['batch'], round(d['decode_tok_s'], 1), 'tok/s', round(d['decode_ms_per_step'], 4), 'ms')"
done; done
