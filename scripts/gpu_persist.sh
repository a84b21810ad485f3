#!/bin/bash
# persistent skinny kernel: parity subset + A/B timing against the one-block-per-workgroup launch
cd /root/repo
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_gemm_gpu.py -x -q -m gpu 2>&1 | tail -5
NP=$((1<<21)); P1=$((1<<22)); P3=$((3<<22)); P4=$((4<<22))
for M in 1 4 8 16; do
for shape in "4096 22016" "4096 12288" "11008 4096" "4096 4096" "8192 28672" "8192 10240"; do
  set -- $shape
  bash tools/sweep_variants.sh $M $1 $2 "0 $NP $P1 $P3 $P4"
done; done
python bench_decode.py --model llama2-7b --bs 1 8 2>&1 | tail -2
} > gpurun_out/persist.log 2>&1
tail -100 gpurun_out/persist.log
