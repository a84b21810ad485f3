#!/bin/bash
# where do the 0.3-0.8 us go that the give-up protocol costs the exchange launches?  r03 library / this tree / no sender count (unsafe, timing only) /
# poll limit by count instead of s_memrealtime / both -- forced xk and xw kernels, one session, two rounds
cd ${GRAFT_REPO_ROOT:-$(pwd)}
out=gpurun_out/r04c; mkdir -p $out
SH=128x4096x4096,64x11008x4096,64x4096x6144,64x4096x12288,256x4096x4096
V=xk64=0x24
SHW=512x4096x4096,256x4096x4096,512x11008x4096
VW=xw41s2=0x1205,xw42s4=0x405
{
for r in 1 2; do
  for lib in ab_r03 ab_head libquick_amd; do
    echo "== $lib (round $r)"
    QUICK_AMD_LIB_OVERRIDE=$PWD/quick_amd/lib/$lib.so timeout 300 python tools/wide_probe.py --shapes $SH --variants $V --iters 60 2>&1 | grep "us "
    [ $lib != ab_r03 ] && QUICK_AMD_LIB_OVERRIDE=$PWD/quick_amd/lib/$lib.so timeout 300 python tools/wide_probe.py --shapes $SHW --variants $VW --iters 60 2>&1 | grep "us "
  done
done
} > $out/ab_exch.txt 2>&1
python - <<'P'
import re,collections
cur=None; d=collections.defaultdict(lambda: collections.defaultdict(list))
for l in open('gpurun_out/r04c/ab_exch.txt'):
    m=re.match(r'== (\S+)',l)
    if m: cur=m.group(1); continue
    m=re.match(r'\s*(\S+)\s+(\S+):\s+([\d.]+) us',l)
    if m: d[(m.group(1),m.group(2))][cur].append(float(m.group(3)))
libs=["ab_r03","ab_head","libquick_amd"]
print('shape variant '+' '.join(libs))
for k,v in d.items(): print(k[0],k[1],' '.join('%7.2f'%(sum(v[l])/len(v[l])) if v.get(l) else '   -   ' for l in libs))
P
