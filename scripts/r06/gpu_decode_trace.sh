#!/bin/bash
# what a bs = 64 / bs = 16 decode step is made of on the r06 tree: rocprofv3 --kernel-trace --stats of bench_decode.py
cd "${GRAFT_REPO_ROOT:-$(pwd)}" || exit 1
root=$(pwd); out=$root/gpurun_out/r06/decode_trace; mkdir -p $out; export TMPDIR=/tmp
for spec in "llama2-7b 64" "mistral-7b 64" "llama2-70b 16" "llama2-7b 1"; do
  set -- $spec
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/$1_$2 -o t -- python $root/bench_decode.py --model $1 --bs $2 > $out/$1_$2.json 2> $out/$1_$2.err)
  f=$(find $out/$1_$2 -name '*kernel_stats.csv' | head -1)
  echo "== $1 bs=$2: $(grep -o '"decode_tok_s": [0-9.]*' $out/$1_$2.json | tail -1)"
  python - "$f" <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:14]:
    print(f'   {r["Name"][:120]:120s} calls {int(r["Calls"]):6d} avg {float(r["AverageNs"]) / 1e3:8.2f} us {100 * float(r["TotalDurationNs"]) / tot:5.1f} %')
P
  find $out/$1_$2 -type f ! -name '*kernel_stats.csv' -delete 2>/dev/null
done
