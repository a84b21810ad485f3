#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
bash tools/prof_passes.sh w8x2_m4096_k8192 --M 4096 --K 8192 --N 8192 --kernel 643 --iters 10 --sets 10 > /dev/null 2>&1
bash tools/prof_passes.sh w2x1nr_m512 --M 512 --kernel 4387 --iters 12 --sets 8 > /dev/null 2>&1
bash tools/prof_passes.sh w4x2nr_m4096_k8192 --M 4096 --K 8192 --N 8192 --kernel 4707 --iters 10 --sets 10 > /dev/null 2>&1
for t in w8x2_m4096_k8192 w2x1nr_m512 w4x2nr_m4096_k8192; do echo "##### $t"; cat gpurun_out/pmc_$t/summary.txt; done
