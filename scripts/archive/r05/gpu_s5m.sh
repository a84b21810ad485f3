#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/s5m; mkdir -p $out
{
for m in 1 2; do QUICK_AMD_ATTN_MFMA=$m timeout 900 python -m pytest tests/test_gemm_gpu.py -q -x -k "attention or decode" 2>&1 | tail -1; done
for rep in 1 2; do for m in 1 2; do
  echo "== QUICK_AMD_ATTN_MFMA=$m (round $rep)"
  QUICK_AMD_ATTN_MFMA=$m timeout 600 python tools/time_attention.py 16x64x8 24x64x8 2>&1 | grep -v amdgpu.ids
  QUICK_AMD_ATTN_MFMA=$m timeout 900 python bench_decode.py --model llama2-70b --bs 16 2>&1 | grep -o "\"batch\": [0-9]*\|\"decode_tok_s\": [0-9.]*" | paste -sd' '
done; done
} 2>&1 | tee $out/attn70.txt
