#!/bin/bash
# A/B of two builds inside one GPU session (box-to-box noise cancels): bash scripts/gpu_ab_probe.sh "<shapes>" [rounds] ["variants"]
# variants: quick_amd/lib/ab_old.so (a copy of the previous in-tree build) against the in-tree library
cd "$GRAFT_REPO_ROOT" || exit 1
shapes=${1:-512x4096x4096,1024x4096x4096,2048x4096x4096,4096x4096x4096,4096x4096x11008,4096x11008x4096,8192x8192x8192}
for r in $(seq ${2:-2}); do
  for v in old new; do
    if [ $v = old ]; then export QUICK_AMD_LIB_OVERRIDE=$PWD/quick_amd/lib/ab_old.so; else unset QUICK_AMD_LIB_OVERRIDE; fi
    python tools/wide_probe.py --shapes $shapes --variants "${3:-auto=0}" --iters 30 2>&1 | grep -v amdgpu.ids | cut -c1-78 | sed "s/^/$v /"
  done
done
