#!/bin/bash
# usage: tools/sweep_variants.sh M K N "kid1 kid2 ..."   -> kernel-only and step times per kernel id
M=$1; K=$2; N=$3; shift 3
for kid in $*; do
  printf "M=%s K=%s N=%s kid=%-7s " $M $K $N $kid
  python bench.py --steps 40 --warmup 5 --cpu-seconds 0 --layers "" --K $K --N $N --sweep $M --M $M --sets 8 --kernel $kid 2>&1 >/dev/null | grep "M=" | sed 's/roofline.*//'
done
