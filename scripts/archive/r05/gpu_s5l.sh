#!/bin/bash
# straight-line eight-tile fragment kernel (w4a16_frag8_kernel): parity, then QUICK_AMD_FRAG8=0 / 1 alternated on the Llama-2-70B M = 16 shapes and in the decode bench
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/s5l; mkdir -p $out
{
timeout 900 python -m pytest tests/test_gemm_gpu.py -q -x -k "eight_tile" 2>&1 | tail -4
for rep in 1 2; do for v in 0 1; do
  echo "== QUICK_AMD_FRAG8=$v (round $rep)"
  QUICK_AMD_FRAG8=$v timeout 600 python tools/lean_check.py --no-check --planner-only 16x8192x57344 16x28672x8192 16x8192x8192 12x8192x57344 9x8192x57344 16x8192x10240 2>&1 | grep -v amdgpu.ids | sed 's/planner \[\([a-z]* [a-z=0-9]*\) [^]]*\]/\1/'
done; done
for v in 0 1 0 1; do
  echo "== decode, QUICK_AMD_FRAG8=$v"
  QUICK_AMD_FRAG8=$v timeout 900 python bench_decode.py --model llama2-70b --bs 12 16 2>&1 | grep -o "\"batch\": [0-9]*\|\"decode_tok_s\": [0-9.]*" | paste -sd' '
done
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -3
} 2>&1 | tee $out/frag8.txt
