#!/bin/bash
# kernel trace of the decode harness: what a step is made of besides the GEMMs
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/dtrace
for bs in 1 64; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/dtrace/bs$bs -o t -- python bench_decode.py --model llama2-7b --bs $bs > gpurun_out/dtrace/bs$bs.log 2>&1
  f=$(find gpurun_out/dtrace/bs$bs -name "*kernel_stats.csv" | head -1)
  echo "== bs=$bs"; cat gpurun_out/dtrace/bs$bs.log | tail -1 | cut -c1-300
  python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:14]:
    print(f'{r["Name"][:110]:110s} calls {int(r["Calls"]):6d} avg {float(r["AverageNs"])/1e3:8.2f} us  {float(r["TotalDurationNs"])/tot*100:5.1f}%')
PY
  cp "$f" gpurun_out/dtrace/decode_bs${bs}_kernel_stats.csv
  find gpurun_out/dtrace/bs$bs -name "*.db" -delete; find gpurun_out/dtrace/bs$bs -name "*trace.csv" -delete
done
