#!/bin/bash
# evidence run: the whole -m gpu suite, smoke, the bench line (+ decode rows)
mkdir -p gpurun_out/r06
tag=${1:-full}
timeout 2400 python -m pytest tests -q -m gpu -x > gpurun_out/r06/pytest_gpu_$tag.txt 2>&1; tail -3 gpurun_out/r06/pytest_gpu_$tag.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r06/smoke_$tag.txt 2>&1; tail -1 gpurun_out/r06/smoke_$tag.txt
timeout 1200 python bench.py > gpurun_out/r06/bench_$tag.json 2> gpurun_out/r06/bench_$tag.err; python - <<PY
import json
d=json.loads(open("gpurun_out/r06/bench_$tag.json").read().strip().splitlines()[-1])
print("value",d["value"],d["unit"],"ms_per_step",d["ms_per_step"],"frac",d["roofline"]["frac"])
for r in d.get("sweep",[]): print("  M",r["M"],r.get("kernel_us"),r.get("plan","")[:60],r.get("roofline",{}).get("frac"))
for r in d.get("decode_layers",[]): print("  layer",r.get("M"),r.get("K"),r.get("N"),r.get("kernel_us"),r.get("plan","")[:50],r.get("roofline",{}).get("frac"))
for r in d.get("decode",[]): print("  decode",r["model"],r["batch"],round(r["tok_s"],1))
PY
