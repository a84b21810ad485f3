#!/bin/bash
# pipelined per-head (MHA) attention against the one-register-set kernel (tools/bin/ab_attnold.so = HEAD's decode_ops), alternated; tests
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/s5d; mkdir -p $out
{
timeout 900 python -m pytest tests/test_gemm_gpu.py -q -x -k "attention or decode" 2>&1 | tail -3
QUICK_AMD_ATTN_WAVES=8 timeout 900 python -m pytest tests/test_gemm_gpu.py -q -x -k "rope_attention" 2>&1 | tail -1
for rep in 1 2; do
  for lib in new old; do
    [ $lib = old ] && export QUICK_AMD_LIB_OVERRIDE=$PWD/tools/bin/ab_attnold.so || unset QUICK_AMD_LIB_OVERRIDE
    echo "== $lib (round $rep)"
    timeout 600 python tools/time_attention.py 1x32x32 4x32x32 8x32x32 16x32x32 64x32x32 2>&1 | grep -v amdgpu.ids
  done
done
for lib in new old new old; do
  [ $lib = old ] && export QUICK_AMD_LIB_OVERRIDE=$PWD/tools/bin/ab_attnold.so || unset QUICK_AMD_LIB_OVERRIDE
  echo "== decode, $lib"
  timeout 900 python bench_decode.py --model llama2-7b --bs 1 16 64 2>&1 | grep -v amdgpu.ids | cut -c1-200
done
unset QUICK_AMD_LIB_OVERRIDE
} 2>&1 | tee $out/attention_mha.txt
