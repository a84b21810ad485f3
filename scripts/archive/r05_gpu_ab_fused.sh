#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05
out=gpurun_out/r05/ab_fused.txt; : > $out
for r in 1 2; do for lean in 0 1; do
  echo "== QUICK_AMD_LEAN=$lean (round $r)" >> $out
  QUICK_AMD_LEAN=$lean timeout 300 python tools/time_ops.py 1 2>/dev/null | grep "gemm" >> $out
done; done
cat $out
