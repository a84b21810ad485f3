#!/bin/bash
# correctness then A/B of mid-token loop variants named in XM_VARIANTS (tools/bin/ab_xm_<name>.so)
mkdir -p gpurun_out/r06
tag=${XM_TAG:-jit}
for n in $XM_VARIANTS; do
  [ $n = base ] && continue
  QUICK_AMD_LIB_OVERRIDE=$PWD/tools/bin/ab_xm_$n.so timeout 600 python tools/xm_check.py --no-time 64x4096x4096 33x4096x4096 17x1024x256 50x1536x4096 40x11008x512 > gpurun_out/r06/xm_check_$n.txt 2>&1
  echo "$n wrong: $(grep -c WRONG gpurun_out/r06/xm_check_$n.txt) $(tail -1 gpurun_out/r06/xm_check_$n.txt)"
done
XM_TAG=$tag bash scripts/r06/gpu_xm_ab.sh
