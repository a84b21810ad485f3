#!/bin/bash
# eight channel tiles per workgroup in the fragment flavour of the skinny kernel (x traffic from L2 half of the weights' instead of equal)
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/s5h; mkdir -p $out
timeout 1200 python tools/skinny_ntw.py 2>&1 | grep -v amdgpu.ids | tee $out/skinny_ntw.txt
