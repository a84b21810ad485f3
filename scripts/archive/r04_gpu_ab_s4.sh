#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
out=gpurun_out/r04c; mkdir -p $out
{
for r in 1 2 3; do
  for lib in ab_r03 libquick_amd; do
    echo "== $lib (round $r)"
    QUICK_AMD_LIB_OVERRIDE=$PWD/quick_amd/lib/$lib.so timeout 300 python tools/wide_probe.py --shapes 128x4096x4096,64x4096x6144,64x4096x12288,256x4096x4096 --variants xk64=0x24 --iters 60 2>&1 | grep "us "
    QUICK_AMD_LIB_OVERRIDE=$PWD/quick_amd/lib/$lib.so timeout 300 python tools/wide_probe.py --shapes 256x4096x4096,512x11008x4096 --variants xk128=0x44 --iters 60 2>&1 | grep "us "
  done
done
} > $out/ab_s4.txt 2>&1
python - <<'P'
import re,collections
cur=None; d=collections.defaultdict(lambda: collections.defaultdict(list))
for l in open('gpurun_out/r04c/ab_s4.txt'):
    m=re.match(r'== (\S+)',l)
    if m: cur=m.group(1); continue
    m=re.match(r'\s*(\S+)\s+(\S+):\s+([\d.]+) us.*(slices=\d)',l)
    if m: d[(m.group(1),m.group(2),m.group(4))][cur].append(float(m.group(3)))
for k,v in d.items(): print(k, ' '.join('%s %.2f'%(l,min(v[l])) for l in ('ab_r03','libquick_amd')))
P
