#!/bin/bash
# one sample of the driver's protocol on whatever box this call gets (GEMM legs only): appended to gpurun_out/r04b/box_samples.jsonl
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out/r04b
python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --decode-seconds 0 --layers "" --prefill-layers 4096x4096x4096 2>gpurun_out/r04b/box_err.log | tail -1 > gpurun_out/r04b/box_$1.json
python - "$1" <<'P'
import json,sys
d=json.load(open(f"gpurun_out/r04b/box_{sys.argv[1]}.json"))
r=d["roofline"]
print(json.dumps({"sample":sys.argv[1],"ms_per_step":d["ms_per_step"],"kernel_us":r["kernel_us"],"frac":r["frac"],"frac_inkernel":r.get("frac_inkernel"),
  "sweep_kernel_us":{s["M"]:s["roofline"]["kernel_us"] for s in d["sweep"]},"prefill_4096":[p["kernel_us"] for p in d.get("prefill_layers",[])]}))
P
