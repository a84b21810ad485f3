#!/bin/bash
# r06: the planner at 65..128 tokens against forced launches of every family that can run there (the r05 audit started at 96 tokens; 65..95 had never
# been audited).   bash scripts/r06/gpu_audit_65_95.sh [tag]
cd "${GRAFT_REPO_ROOT:-$(pwd)}" || exit 1
tag=${1:-a}
mkdir -p gpurun_out/r06
LAYERS="4096x4096 4096x6144 4096x8192 8192x4096 4096x12288 4096x22016 11008x4096 4096x28672 14336x4096 8192x8192 8192x10240 28672x8192 5120x5120 5120x13824 13824x5120 4096x14336"
T=""
for kn in $LAYERS; do
  k=${kn%x*}; n=${kn#*x}
  for m in 65 72 80 88 95 96 112 128; do T="$T,${m}x${k}x${n}"; done
done
T=${T#,}
VT="auto=0,auto2=0,xw21s1=0x125,xw21s2=0x225,xw41s1=0x1145,xw41s2=0x1245,xw41s4=0x1445,xw42s2=0x245,xw42s4=0x445,xk2=0x24,xk4=0x44,tiled=2,xm11=0x117,xm12=0x127,xm13=0x137,xm21=0x217,xm22=0x227,xm23=0x237"
timeout 2400 python tools/wide_probe.py --shapes $T --variants $VT --iters 20 --out gpurun_out/r06/audit_65_128_$tag.jsonl > gpurun_out/r06/audit_65_128_$tag.log 2>&1
tail -n 2 gpurun_out/r06/audit_65_128_$tag.log
