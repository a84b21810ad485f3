#!/bin/bash
# r05 planner audit: the planner's pick against forced launches of every family on 13 layer shapes x 18 token counts, ONE session
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05
S=""
for kn in 4096x4096 4096x6144 4096x12288 4096x22016 11008x4096 4096x28672 14336x4096 8192x8192 8192x10240 28672x8192 5120x5120 5120x13824 13824x5120; do
  k=${kn%x*}; n=${kn#*x}
  for m in 33 48 64 96 128 160 192 256 320 384 512 640 768 1024 1536 2048 3072 4096; do S="$S,${m}x${k}x${n}"; done
done
S=${S#,}
V="auto=0,auto2=0,xw21s1=0x125,xw21s2=0x225,xw41s1=0x1145,xw41s2=0x1245,xw41s4=0x1445,xw42s1=0x145,xw42s2=0x245,xw42s4=0x445,xw82=0x185,xk2=0x24,xk4=0x44,wide=3,tiled=2"
timeout 3000 python tools/wide_probe.py --shapes $S --variants $V --iters 20 --out gpurun_out/r05/planner_audit.jsonl > gpurun_out/r05/planner_audit.log 2>&1
tail -3 gpurun_out/r05/planner_audit.log
