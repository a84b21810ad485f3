#!/bin/bash
# per-kernel breakdown of decode steps (rocprofv3 kernel trace of bench_decode.py):  bash scripts/gpu_decode_prof.sh <model> <bs>
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root; mkdir -p gpurun_out/decode_prof
out=$root/gpurun_out/decode_prof
export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o dec -- python $root/bench_decode.py --model ${1:-llama2-7b} --bs ${2:-1} --gen 64 > $out/dec.json 2> $out/dec.err)
tail -1 $out/dec.json | cut -c1-200
python - <<PY
import csv, glob
f = glob.glob("$out/**/dec_kernel_stats.csv", recursive=True)
if f:
    for i, r in enumerate(csv.DictReader(open(f[0]))):
        if i < 16: print(r["Name"][:90].ljust(90), r["Calls"].rjust(7), r["AverageNs"].rjust(10), r["Percentage"].rjust(7))
PY
find $out -type f ! -name '*kernel_stats.csv' ! -name 'dec.json' -delete
