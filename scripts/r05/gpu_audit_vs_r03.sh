#!/bin/bash
# r05 audit: the planner's pick of this tree against r03's library (tools/bin/ab_r03.so, built from commit 05601e8) in ONE session, alternating,
# on the layer shapes of the three BASELINE models + Llama-2-13B, 33 .. 4096 tokens (the shape set of scripts/archive/r04_gpu_audit_vs_r03.sh)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05v3
S=""
for kn in 4096x4096 4096x6144 4096x12288 4096x22016 11008x4096 4096x28672 14336x4096 8192x8192 8192x10240 28672x8192 5120x5120 5120x13824 13824x5120; do
  k=${kn%x*}; n=${kn#*x}
  for m in 33 48 64 96 128 160 192 256 320 384 512 640 768 1024 1536 2048 3072 4096; do S="$S,${m}x${k}x${n}"; done
done
S=${S#,}
rm -f gpurun_out/r05v3/audit_*.jsonl
for r in 1 2; do
QUICK_AMD_LIB_OVERRIDE=$PWD/tools/bin/ab_r03.so timeout 1200 python tools/wide_probe.py --shapes $S --variants auto=0 --iters 24 --out gpurun_out/r05v3/audit_r03_$r.jsonl > /dev/null 2>&1
timeout 1200 python tools/wide_probe.py --shapes $S --variants auto=0 --iters 24 --out gpurun_out/r05v3/audit_new_$r.jsonl > /dev/null 2>&1
done
python tools/audit_vs_r03.py gpurun_out/r05v3 | sed 's/ r04 / r05 /' > gpurun_out/r05v3/audit_vs_r03.txt
tail -1 gpurun_out/r05v3/audit_vs_r03.txt; grep -c "slower" gpurun_out/r05v3/audit_vs_r03.txt
