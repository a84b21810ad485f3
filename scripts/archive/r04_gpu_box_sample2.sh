#!/bin/bash
# driver protocol (20 steps, 5 warm-up) against the default (2000 / 50) on ONE box, alternated
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out/r04b
for rep in 1 2 3; do
for proto in "20 5" "2000 50"; do
  set -- $proto
  python bench.py --steps $1 --warmup $2 --cpu-seconds 0 --decode-seconds 0 --layers "" --prefill-layers "" --sweep 512 2>>gpurun_out/r04b/box_err.log | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(json.dumps({'steps':d['steps'],'ms_per_step':round(d['ms_per_step']*1e3,2),'kernel_us':round(r['kernel_us'],2),'kernel_us_event_pairs':round(r['kernel_us_event_pairs'],2),'inkernel':r.get('kernel_us_inkernel'),'frac':round(r['frac'],3)}))"
done
done | tee gpurun_out/r04b/protocols.txt
