#!/bin/bash
# A/B: weight requests of the lean kernels with the nt (streaming) cache policy against the default, alternating in one session
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05
out=gpurun_out/r05/ab_lean_nt.txt; : > $out
for r in 1 2; do for v in lean_base lean_nt; do
  echo "== $v (round $r)" >> $out
  QUICK_AMD_LIB_OVERRIDE=$PWD/tools/bin/ab_$v.so timeout 600 python tools/lean_check.py --no-check --planner-only 1x4096x4096 1x4096x12288 1x4096x22016 1x11008x4096 4x4096x22016 16x4096x4096 16x4096x22016 2>&1 | grep -v amdgpu | sed 's/planner \[lean //; s/tiles_per_wave.*workspace=0\]//' >> $out
  QUICK_AMD_LIB_OVERRIDE=$PWD/tools/bin/ab_$v.so timeout 600 python bench_decode.py --model llama2-7b --bs 1 16 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    print('decode', d['model'], 'bs', d['batch'], round(d['decode_tok_s'], 1), 'tok/s', round(d['decode_ms_per_step'], 4), 'ms')" >> $out
done; done
cat $out
