#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee gpurun_out/r03_tests.txt
B="python bench.py --steps 20 --warmup 5 --sweep 512 --layers= --prefill-layers= --cpu-seconds 0 --decode-seconds 0"
for rep in 1 2 3; do
  for k in 0 $((3+32+256+(1<<15)+(4<<22))) $((4+(4<<4))); do
    $B --kernel $k 2>&1 >/dev/null | grep "M= 512" | sed "s/^/kernel=$k /"
  done
done | tee gpurun_out/r03_ab_m512.txt
