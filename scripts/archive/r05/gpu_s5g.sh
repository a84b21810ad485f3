#!/bin/bash
# four-tile fragment flavour of the skinny kernel (the M = 16 launches of Llama-2-70B: 34-54 % of HBM, one 8-wave workgroup per CU, one
# 4 KiB chunk per wave ahead): A/B builds -- u2 = two k tiles per chunk (242 registers), lb2 = capped at 128 registers, two workgroups per CU
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/s5g; mkdir -p $out
{
QUICK_AMD_ATTN_MFMA=1 timeout 900 python -m pytest tests/test_gemm_gpu.py -q -x -k "attention or decode" 2>&1 | tail -2
for rep in 1 2; do
for v in base u2 lb2; do
  [ $v = base ] && unset QUICK_AMD_LIB_OVERRIDE || export QUICK_AMD_LIB_OVERRIDE=$PWD/tools/bin/ab_$v.so
  echo "== $v (round $rep)"
  timeout 600 python tools/lean_check.py --no-check --planner-only 16x8192x10240 16x8192x8192 16x8192x57344 16x28672x8192 64x4096x4096 32x4096x8192 12x8192x57344 2>&1 | grep -v amdgpu.ids | sed 's/planner \[\([a-z]*\) [^]]*\]/\1/'
done; done
for v in base u2 lb2 base u2 lb2; do
  [ $v = base ] && unset QUICK_AMD_LIB_OVERRIDE || export QUICK_AMD_LIB_OVERRIDE=$PWD/tools/bin/ab_$v.so
  echo "== decode, $v"
  timeout 900 python bench_decode.py --model llama2-70b --bs 16 2>&1 | grep -o "\"batch\": [0-9]*\|\"decode_tok_s\": [0-9.]*" | paste -sd' '
  timeout 900 python bench_decode.py --model llama2-7b --bs 64 2>&1 | grep -o "\"batch\": [0-9]*\|\"decode_tok_s\": [0-9.]*" | paste -sd' '
done
unset QUICK_AMD_LIB_OVERRIDE
for v in u2 lb2; do echo "-- tests, $v"; QUICK_AMD_LIB_OVERRIDE=$PWD/tools/bin/ab_$v.so timeout 1200 python -m pytest tests/test_gemm_gpu.py -q -x -k "skinny or baseline or synthetic or deferred or layer_shapes or decode" 2>&1 | tail -2; done
} 2>&1 | tee $out/skinny_variants.txt
