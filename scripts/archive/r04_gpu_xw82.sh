#!/bin/bash
# the generated 256 x 256 loop (xw tokens=256: waves of 256 x 64, ring of two slots) against r02's hipcc-scheduled 256 x 256 kernel and the
# generated 128 x 256 loop: correctness first, then one-session timings on the prefill shapes
cd ${GRAFT_REPO_ROOT:-$(pwd)}
out=gpurun_out/r04b; mkdir -p $out
timeout 600 python tools/xw_check.py 2>&1 | grep -v amdgpu.ids | grep "tokens=256\|FAIL" > $out/xw82_check.txt; tail -3 $out/xw82_check.txt
S=""
for kn in 4096x4096 4096x11008 4096x12288 11008x4096 8192x8192 4096x22016 8192x10240 28672x8192 4096x14336; do
  k=${kn%x*}; n=${kn#*x}
  for m in 1024 2048 4096 8192; do S="$S,${m}x${k}x${n}"; done
done
S=${S#,}
timeout 1500 python tools/wide_probe.py --shapes $S --variants auto=0,wide256=0x283,xw256=0x85,xw42s1=0x105 --iters 30 --out $out/xw82_sweep.jsonl > $out/xw82_sweep.txt 2>&1
tail -3 $out/xw82_sweep.txt
python - <<'P'
import json,collections
rows=[json.loads(l) for l in open('gpurun_out/r04b/xw82_sweep.jsonl')]
by=collections.defaultdict(dict)
for r in rows: by[r['shape']][r['variant']]=r['kernel_us']
import math
g=[]
for sh,d in by.items():
    if 'xw256' in d and 'wide256' in d:
        g.append(d['xw256']/min(d['auto'],d['wide256']))
        print(sh, ' '.join(f"{k}:{v:.1f}" for k,v in d.items()), f"xw256/best-other {g[-1]:.3f}")
print('geomean', math.exp(sum(map(math.log,g))/len(g)))
P
