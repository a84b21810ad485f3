#!/bin/bash
# skinny kernel, fragments-from-L2 loop: the next chunk requested unconditionally (product) against r01-r04's guarded requests (tools/bin/ab_guard.so),
# whose prefetch hipcc drained in front of the compute; parity suite on the product; eight-tile flavour debug
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/s5i; mkdir -p $out
{
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -3
python tools/_ntw8_dbg.py 2>&1 | grep -v amdgpu.ids
for rep in 1 2; do
for v in new guard; do
  [ $v = new ] && unset QUICK_AMD_LIB_OVERRIDE || export QUICK_AMD_LIB_OVERRIDE=$PWD/tools/bin/ab_$v.so
  echo "== $v (round $rep)"
  timeout 600 python tools/lean_check.py --no-check --planner-only 16x8192x10240 16x8192x8192 16x8192x57344 16x28672x8192 64x4096x4096 32x4096x8192 12x8192x57344 8x8192x10240 48x4096x4096 24x4096x12288 6x4096x12288 16x4096x6144 2>&1 | grep -v amdgpu.ids | sed 's/planner \[\([a-z]* [a-z=0-9]*\) [^]]*\]/\1/'
done; done
for v in new guard new guard; do
  [ $v = new ] && unset QUICK_AMD_LIB_OVERRIDE || export QUICK_AMD_LIB_OVERRIDE=$PWD/tools/bin/ab_$v.so
  echo "== decode, $v"
  timeout 900 python bench_decode.py --model llama2-70b --bs 16 2>&1 | grep -o "\"batch\": [0-9]*\|\"decode_tok_s\": [0-9.]*" | paste -sd' '
  timeout 900 python bench_decode.py --model llama2-7b mistral-7b --bs 16 64 2>&1 | grep -o "\"batch\": [0-9]*\|\"decode_tok_s\": [0-9.]*" | paste -sd' '
done
unset QUICK_AMD_LIB_OVERRIDE
} 2>&1 | tee $out/skinny_uncond.txt
