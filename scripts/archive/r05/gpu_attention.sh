#!/bin/bash
# r05: the single-query attention kernel, WAVES x UNR forms with early cache requests, one session
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05
{
timeout 900 python -m pytest tests/test_gemm_gpu.py -q -x -k "rope_attention or decode" 2>&1 | tail -3
for w in 4 8; do QUICK_AMD_ATTN_WAVES=$w timeout 900 python -m pytest tests/test_gemm_gpu.py -q -x -k "rope_attention" 2>&1 | tail -1; done
for rep in 1 2; do
  for w in 4 8; do
    echo "== QUICK_AMD_ATTN_WAVES=$w (round $rep)"
    QUICK_AMD_ATTN_WAVES=$w timeout 600 python tools/time_attention.py 1x32x32 4x32x32 8x32x32 64x32x32 2>&1 | grep -v amdgpu.ids
  done
done
for w in 4 8; do
  echo "== decode, QUICK_AMD_ATTN_WAVES=$w"
  QUICK_AMD_ATTN_WAVES=$w timeout 900 python bench_decode.py --model llama2-7b --bs 1 8 64 2>&1 | grep -v amdgpu.ids | cut -c1-260
done
} > gpurun_out/r05/attention.txt 2>&1
cat gpurun_out/r05/attention.txt
