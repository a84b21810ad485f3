#!/bin/bash
# Llama-2-70B decode: (a) RMSNorm prologue of the fragment flavour un-fused on large layers (QUICK_AMD_LN_FRAGMENT_MAX), (b) the 8-head MFMA attention from 128 pairs
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/s5j; mkdir -p $out
{
for rep in 1 2; do
for cfg in "base" "ln0 QUICK_AMD_LN_FRAGMENT_MAX=0" "ln100M QUICK_AMD_LN_FRAGMENT_MAX=100000000" "attn128 QUICK_AMD_ATTN_MFMA8_MIN_PAIRS=128" "both QUICK_AMD_LN_FRAGMENT_MAX=100000000 QUICK_AMD_ATTN_MFMA8_MIN_PAIRS=128"; do
  set -- $cfg; name=$1; shift
  echo "== $name (round $rep)"
  env "$@" timeout 900 python bench_decode.py --model llama2-70b --bs 8 16 32 2>&1 | grep -o "\"batch\": [0-9]*\|\"decode_tok_s\": [0-9.]*" | paste -sd' '
done; done
for cfg in "base" "ln0 QUICK_AMD_LN_FRAGMENT_MAX=0"; do
  set -- $cfg; name=$1; shift
  echo "== 7B, $name"
  env "$@" timeout 900 python bench_decode.py --model llama2-7b mistral-7b --bs 8 16 32 2>&1 | grep -o "\"batch\": [0-9]*\|\"decode_tok_s\": [0-9.]*" | paste -sd' '
done
QUICK_AMD_ATTN_MFMA8_MIN_PAIRS=128 timeout 600 python -m pytest tests/test_gemm_gpu.py -q -x -k "attention" 2>&1 | tail -1
} 2>&1 | tee $out/decode70_ab.txt
