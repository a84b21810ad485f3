#!/bin/bash
# grouped-query attention with the scores on the matrix core (QUICK_AMD_ATTN_MFMA: 0 = vector-ALU sweep, 1 = one 16-row chunk per register
# set, 2 = two), alternated in one session; tests under every setting
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/s5e; mkdir -p $out
{
for m in 1 2 0; do echo "-- tests, QUICK_AMD_ATTN_MFMA=$m"; QUICK_AMD_ATTN_MFMA=$m timeout 900 python -m pytest tests/test_gemm_gpu.py -q -x -k "attention or decode" 2>&1 | tail -3; done
for rep in 1 2; do
  for m in 0 1 2; do
    echo "== QUICK_AMD_ATTN_MFMA=$m (round $rep)"
    QUICK_AMD_ATTN_MFMA=$m timeout 600 python tools/time_attention.py 64x32x8 32x32x8 64x64x8 32x64x8 2>&1 | grep -v amdgpu.ids
  done
done
for m in 0 1 2 0 1 2; do
  echo "== decode, QUICK_AMD_ATTN_MFMA=$m"
  QUICK_AMD_ATTN_MFMA=$m timeout 900 python bench_decode.py --model mistral-7b --bs 32 64 2>&1 | grep -o "\"batch\": [0-9]*\|\"decode_tok_s\": [0-9.]*" | paste -sd' '
done
for m in 0 1 0 1; do
  echo "== decode 70B, QUICK_AMD_ATTN_MFMA=$m"
  QUICK_AMD_ATTN_MFMA=$m timeout 900 python bench_decode.py --model llama2-70b --bs 32 2>&1 | grep -o "\"batch\": [0-9]*\|\"decode_tok_s\": [0-9.]*" | paste -sd' '
done
} 2>&1 | tee $out/attention_gqa_mfma.txt
