#!/bin/bash
# skinny kernel variants: parity + A/B timing
cd /root/repo
mkdir -p gpurun_out
{
timeout 1200 python -m pytest tests/test_gemm_gpu.py -x -q -m gpu 2>&1 | tail -5
W4=$((1<<8)); W4C4=$(((1<<8)|(4<<22))); W4C3=$(((1<<8)|(3<<22))); W16=$((4<<8)); W16C1=$(((4<<8)|(1<<22)))
for M in 1 8; do
for shape in "4096 22016" "4096 12288" "11008 4096" "4096 4096" "8192 28672"; do
  set -- $shape
  bash tools/sweep_variants.sh $M $1 $2 "0 $W4 $W4C4 $W4C3 $W16 $W16C1"
done; done
python bench_decode.py --model llama2-7b --bs 1 8
} > gpurun_out/persist.log 2>&1
tail -80 gpurun_out/persist.log
