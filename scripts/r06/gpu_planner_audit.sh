#!/bin/bash
# r06 planner audit: the planner's pick against forced launches of every family on 13 layer shapes x 18 token counts, ONE session (r05's set; 33..128 tokens also against
# every mid-token configuration), then the mid-token audit of tools/xm_audit.py on its 15 layers x 8 token counts.   bash scripts/r06/gpu_planner_audit.sh [tag]
cd "${GRAFT_REPO_ROOT:-$(pwd)}" || exit 1
tag=${1:-a}
mkdir -p gpurun_out/r06
LAYERS="4096x4096 4096x6144 4096x12288 4096x22016 11008x4096 4096x28672 14336x4096 8192x8192 8192x10240 28672x8192 5120x5120 5120x13824 13824x5120"
S=""; T=""
for kn in $LAYERS; do
  k=${kn%x*}; n=${kn#*x}
  for m in 33 48 64 80 96 128; do T="$T,${m}x${k}x${n}"; done
  for m in 96 128 160 192 256 320 384 512 640 768 1024 1536 2048 3072 4096; do S="$S,${m}x${k}x${n}"; done
done
S=${S#,}; T=${T#,}
V="auto=0,auto2=0,xw21s1=0x125,xw21s2=0x225,xw41s1=0x1145,xw41s2=0x1245,xw41s4=0x1445,xw42s1=0x145,xw42s2=0x245,xw42s4=0x445,xw82=0x185,xk2=0x24,xk4=0x44,wide=3,tiled=2"
VT="auto=0,auto2=0,xw21s1=0x125,xw21s2=0x225,xw41s2=0x1245,xw41s4=0x1445,xk2=0x24,xk4=0x44,skinny4=0x41,tiled=2,xm11=0x117,xm12=0x127,xm13=0x137,xm21=0x217,xm22=0x227,xm23=0x237"
timeout 1500 python tools/wide_probe.py --shapes $T --variants $VT --iters 20 --out gpurun_out/r06/planner_audit_mid_$tag.jsonl > gpurun_out/r06/planner_audit_mid_$tag.log 2>&1
timeout 2400 python tools/wide_probe.py --shapes $S --variants $V --iters 20 --out gpurun_out/r06/planner_audit_$tag.jsonl > gpurun_out/r06/planner_audit_$tag.log 2>&1
QUICK_AMD_XM=0 timeout 900 python tools/xm_audit.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06/xm_audit_$tag.txt
tail -2 gpurun_out/r06/planner_audit_$tag.log gpurun_out/r06/planner_audit_mid_$tag.log; tail -2 gpurun_out/r06/xm_audit_$tag.txt | cut -c1-150
