#!/bin/bash
# parity tests + per-kernel breakdown of one decode step (rocprofv3 kernel trace of bench_decode.py)
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root; mkdir -p gpurun_out/decode_prof
out=$root/gpurun_out/decode_prof
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o dec -- python $root/bench_decode.py --model llama2-7b --bs ${1:-1} --gen 64 > $out/dec.json 2> $out/dec.err)
tail -1 $out/dec.json | cut -c1-300
python - <<PY
import csv, glob, collections
f = glob.glob("$out/**/dec_kernel_stats.csv", recursive=True)
if f:
    for i, r in enumerate(csv.DictReader(open(f[0]))):
        if i < 14: print(r["Name"][:70].ljust(70), r["Calls"].rjust(7), r["AverageNs"].rjust(10), r["Percentage"].rjust(7))
PY
