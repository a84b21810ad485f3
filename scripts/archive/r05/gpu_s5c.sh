#!/bin/bash
# (1) pipelined GQA attention against the one-register-set kernel (tools/bin/ab_attnold.so = HEAD's decode_ops), alternated; its tests
# (2) the 256 x 256 tile's first round started apart inside the XCDs (tools/ab_xw_exit.py)
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/s5c; mkdir -p $out
{
timeout 900 python -m pytest tests/test_gemm_gpu.py -q -x -k "attention or decode" 2>&1 | tail -3
for rep in 1 2; do
  for lib in new old; do
    [ $lib = old ] && export QUICK_AMD_LIB_OVERRIDE=$PWD/tools/bin/ab_attnold.so || unset QUICK_AMD_LIB_OVERRIDE
    echo "== $lib (round $rep)"
    timeout 600 python tools/time_attention.py 64x32x8 16x32x8 32x32x8 64x64x8 16x64x8 2>&1 | grep -v amdgpu.ids
  done
done
for lib in new old new old; do
  [ $lib = old ] && export QUICK_AMD_LIB_OVERRIDE=$PWD/tools/bin/ab_attnold.so || unset QUICK_AMD_LIB_OVERRIDE
  echo "== decode, $lib"
  timeout 900 python bench_decode.py --model mistral-7b --bs 16 64 2>&1 | grep -v amdgpu.ids | cut -c1-200
done
unset QUICK_AMD_LIB_OVERRIDE
} 2>&1 | tee $out/attention_gqa.txt
timeout 900 python tools/ab_xw_exit.py 8192x4096x22016 8192x4096x12288 2048x4096x22016 8192x11008x4096 4096x28672x8192 2>&1 | grep -v amdgpu.ids | tee $out/ab_xw_exit2.txt
