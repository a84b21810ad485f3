#!/bin/bash
# r04: where do the four-wave hand-placed kernels beat the planner's pick?  (rows for the planner rules)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04
S=""
for kn in 4096x4096 4096x11008 4096x12288 11008x4096 8192x8192 4096x14336 14336x4096 4096x22016 8192x10240; do
  k=${kn%x*}; n=${kn#*x}
  for m in 96 128 192 256 384 512 768 1024 1536 2048 4096 8192; do S="$S,${m}x${k}x${n}"; done
done
S=${S#,}
timeout 1500 python tools/wide_probe.py --shapes $S --variants auto=0,xw42=0x5,xw41=0x1005,xw21=0x25,xw42s1=0x105,xw41s1=0x1105,xw41s2=0x1205 --iters 30 --out gpurun_out/r04/xw_sweep.jsonl > gpurun_out/r04/xw_sweep.txt 2>&1
tail -5 gpurun_out/r04/xw_sweep.txt
