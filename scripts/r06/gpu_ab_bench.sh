#!/bin/bash
# A/B of whole-library variants (tools/bin/ab_<name>.so) on the bench's sweep and decode-layer rows, alternating, two rounds
mkdir -p gpurun_out/r06
out=gpurun_out/r06/ab_bench_${AB_TAG:-1}.txt; : > $out
for r in 1 2; do
for n in $AB_VARIANTS; do
  if [ $n = base ]; then unset QUICK_AMD_LIB_OVERRIDE; else export QUICK_AMD_LIB_OVERRIDE=$PWD/tools/bin/ab_$n.so; fi
  timeout 600 python bench.py --cpu-seconds 0 --decode-seconds ${AB_DECODE:-0} --prefill-layers "${AB_PREFILL-}" --layers "${AB_LAYERS:-1x4096x12288,1x4096x22016,1x11008x4096,8x4096x12288,16x4096x22016,16x8192x57344}" 2>/dev/null | tail -1 > /tmp/ab_line.json
  python - "$n" "$r" >> $out <<'PY'
import json, sys
d = json.loads(open("/tmp/ab_line.json").read())
rows = [f"M={r['M']}: {r['roofline']['kernel_us']:.2f}" for r in d.get("sweep", [])] + [f"{r['M']}x{r['K']}x{r['N']}: {r['kernel_us']:.2f}" for r in d.get("decode_layers", [])] + \
       [f"{r['M']}x{r['K']}x{r['N']}: {r['kernel_us']:.1f}" for r in d.get("prefill_layers", [])] + [f"{r['model']} bs={r['batch']}: {r['tok_s']:.0f}" for r in d.get("decode", [])]
print(f"{sys.argv[1]:8s} round {sys.argv[2]}: step {d['ms_per_step'] * 1e3:.2f} us | " + " | ".join(rows))
PY
done; done
cat $out
