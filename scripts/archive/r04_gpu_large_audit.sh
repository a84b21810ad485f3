#!/bin/bash
# large-M audit on layer shapes the rules were NOT tuned on (Llama-2-13B, Qwen2-7B, Yi-34B, Llama-3-8B) + odd token counts: the planner's pick against forced selections
cd ${GRAFT_REPO_ROOT:-$(pwd)}
out=gpurun_out/r04c; mkdir -p $out
S=""
for kn in 5120x5120 5120x13824 13824x5120 5120x15360 3584x18944 18944x3584 3584x4608 7168x7168 7168x20480 20480x7168 4096x28672 14336x4096; do
  k=${kn%x*}; n=${kn#*x}
  for m in 768 1536 3072 6144; do S="$S,${m}x${k}x${n}"; done
done
S=${S#,}
timeout 2000 python tools/wide_probe.py --shapes $S --variants auto=0,wide256=0x283,xw256=0x85,xw42s1=0x105,xw42=0x5,xw41s1=0x1105,wide=3 --iters 20 --out $out/large_audit.jsonl > $out/large_audit.txt 2>&1
tail -2 $out/large_audit.txt
python - <<'P'
import json,collections,math
rows=[json.loads(l) for l in open('gpurun_out/r04c/large_audit.jsonl')]
by=collections.defaultdict(dict)
for r in rows: by[r['shape']][r['variant']]=(r['kernel_us'],r['plan'])
g=[]
for sh,d in by.items():
    best=min(v[0] for v in d.values()); a=d['auto'][0]; g.append(a/best)
    flag=' <==' if a/best>1.04 else ''
    print(f"{sh:18s} auto {a:8.1f} [{' '.join(d['auto'][1].split()[:3])}]  best {best:8.1f} ({min(d,key=lambda k:d[k][0])})  gap {a/best:.3f}{flag}")
print('mean gap',sum(g)/len(g),'worst',max(g))
P
