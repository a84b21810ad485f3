#!/bin/bash
# A/B: weight requests of the r01 skinny kernels with the nt cache policy against the default
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05
out=gpurun_out/r05/ab_skinny_nt.txt; : > $out
for r in 1 2; do for v in skinny_base skinny_nt; do
  echo "== $v (round $r)" >> $out
  QUICK_AMD_LIB_OVERRIDE=$PWD/tools/bin/ab_$v.so timeout 900 python tools/wide_probe.py --iters 30 --shapes 16x8192x57344,16x28672x8192,16x8192x10240,16x8192x8192,1x28672x8192,6x4096x22016,64x4096x4096,48x4096x4096,32x4096x8192,64x5120x5120 --variants auto=0 2>&1 | grep auto | cut -c1-120 >> $out
  QUICK_AMD_LIB_OVERRIDE=$PWD/tools/bin/ab_$v.so timeout 900 python bench_decode.py --model llama2-70b --bs 16 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    print('decode', d['model'], 'bs', d['batch'], round(d['decode_tok_s'], 1), 'tok/s', round(d['decode_ms_per_step'], 4), 'ms')" >> $out
done; done
cat $out
